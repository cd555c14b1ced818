def deprecate(*args, **kwargs):
    pass


def is_scipy_available():
    return False
