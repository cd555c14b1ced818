"""CPU oracle (test infrastructure only -- never imported by the product package)."""
