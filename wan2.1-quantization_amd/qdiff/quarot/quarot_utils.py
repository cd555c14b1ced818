"""Hadamard utilities.  The reference carries ~97k lines of literal +-1 tables (quarot_utils.py:269ff); the ones the
Wan shapes need are exactly Paley-I matrices (orders 12, 20, 60, 108, 140) and H2 (x) had20 (order 40), so they are
constructed here from quadratic residues (checked against products of the reference's tables: tests/golden a5_*)."""
import math

import numpy as np
import torch


def is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def paley_hadamard(q):
    """Order q+1, q prime = 3 mod 4; first column +1, first row (+1,-1,...), core chi(i-j) off-diagonal, +1 diagonal."""
    chi = -np.ones(q, dtype=np.int64)
    chi[[(a * a) % q for a in range(1, q)]] = 1
    chi[0] = 0
    i = np.arange(q)
    H = np.ones((q + 1, q + 1), dtype=np.int64)
    H[0, 1:] = -1
    H[1:, 1:] = chi[(i[:, None] - i[None, :]) % q] + np.eye(q, dtype=np.int64)
    return H


def sylvester(n):
    H = np.ones((1, 1), dtype=np.int64)
    while H.shape[0] < n:
        H = np.block([[H, H], [H, -H]])
    return H


_PALEY = {12: 11, 20: 19, 60: 59, 108: 107, 140: 139}


def get_hadK(n, transpose=False, strict=False):
    """(hadK as float64 tensor or None, K), the reference's precedence order (quarot_utils.py:100-155).  Tables not constructible
    here raise NotImplementedError (orders 172, 156, 144, 52, 36, 28 -- none occurs in Wan2.1).
    strict=True is the reference to the letter: the FIRST K that divides n must leave a power-of-two co-factor, else
    AssertionError -- which makes 13824 (Wan2.1-14B ffn.2) unrotatable: 13824 % 144 == 0 trips the assert at :110-112 before the
    K = 108 branch (:118-121) is reached, although 13824 = 108 x 128 fits it.  The default is REPO-DEFINED for exactly
    those sizes: a K with a non-power-of-two co-factor is skipped and the reference's own order continues, so 13824 gets
    K = 108 (the reference's get_had108 table = Paley-107) x H_128.  Every size the reference accepts gives the same (hadK, K)."""
    for K in (172, 156, 144, 140, 108, 60, 52, 36, 28, 40, 20, 12):
        if n % K == 0:
            if not is_pow2(n // K):
                assert not strict, f"cannot build a Hadamard transform of size {n} = {K} x {n // K}"
                continue
            if K in _PALEY:
                H = paley_hadamard(_PALEY[K])
            elif K == 40:
                H = np.kron(np.array([[1, 1], [1, -1]]), paley_hadamard(19))
            else:
                raise NotImplementedError(f"Hadamard table of order {K} is not constructed in this build")
            H = torch.from_numpy(H.T.copy() if transpose else H).to(torch.float64)
            return H, K
    assert is_pow2(n), f"cannot build a Hadamard transform of size {n}"
    return None, 1


def matmul_hadU(X, transpose=False):
    """(hadK (x) H_{n/K}) X / fp32-sqrt(n) along the last axis; any device, dtype of X (use float64 for weights)
    (reference quarot_utils.py:158-179)."""
    n = X.shape[-1]
    hadK, K = get_hadK(n, transpose)
    m = n // K
    v = X.reshape(-1, K, m)
    h = 1
    while h < m:
        v = v.reshape(-1, K, m // (2 * h), 2, h)
        a, b = v[..., 0, :], v[..., 1, :]
        v = torch.stack([a + b, a - b], dim=-2).reshape(-1, K, m)
        h *= 2
    if K > 1:
        v = torch.matmul(hadK.to(device=X.device, dtype=X.dtype), v)
    div = torch.tensor(n).sqrt().item()  # fp32 sqrt, as the reference
    return v.reshape(X.shape) / div


def random_hadamard_signs(size, generator=None):
    return (torch.randint(0, 2, (size,), generator=generator) * 2 - 1).to(torch.float64)


def random_hadamard_matrix(size, device, signs=None):
    """R = hadU(diag(+-1)) in float64 (reference :186-192).  Row i of R is sign_i * hadU(e_i): x @ R == hadU(x * signs)."""
    s = random_hadamard_signs(size) if signs is None else signs.to(torch.float64)
    return matmul_hadU(torch.diag(s)).to(device)


def kernel_rotation_params(n, device):
    """(had_k, hadk fp32 [K',K'] or None) for wanq_rotate_quant_rows, i.e. hadU written as (H_K' (x) H_128)/sqrt(n)
    with K' = K * m/128.  Only had_k crosses the C boundary (the library generates the same table from n); hadk is returned
    for host-side checks.  n = 8960 = 140 x 64 has its own kernel (csrc/rotate140.hip: had_k = 140, 64-wide blocks, no
    LayerNorm form).  None when the library has no kernel for n: any other block size m < 128 or a K' outside
    {2^p <= 32, 12, 40} (csrc/rotate.hip)."""
    hadK, K = get_hadK(n)
    m = n // K
    if n == 8960:
        return 140, hadK.float().contiguous().to(device)
    if n == 13824:
        return 108, hadK.float().contiguous().to(device)
    if m < 128:
        return None
    kk = K * (m // 128)
    if kk not in (1, 2, 4, 8, 16, 32, 12, 40):
        return None
    if kk == 1:
        return 1, None
    H = np.kron(hadK.numpy().astype(np.int64) if hadK is not None else np.ones((1, 1), dtype=np.int64), sylvester(m // 128))
    return kk, torch.from_numpy(H.astype(np.float32)).contiguous().to(device)
