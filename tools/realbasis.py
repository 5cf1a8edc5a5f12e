"""A basis in which the hopping blocks of a collinear operator are REAL (numpy only; the analysis tool behind DESIGN.md section 3,
"a structural halving that was evaluated": every fixture of the reference's cases admits such a basis, and on the GPU the recursion
coefficients of the transformed operator equal U^H A_n U of the original one to 1e-13).

For blocks H_s (9x9 per spin) that are real in SOME orthonormal basis C -- the two-centre spd integrals in real harmonics -- the equations
H_s W = W conj(H_s) have the solution W = C C^T (unitary, symmetric).  find_real_basis solves them in the least-squares sense (smallest
singular vector), checks that W is unitary and symmetric, and returns C from the Takagi factorisation W = C C^T: C^H H_s C is then real."""
import numpy as np


def find_real_basis(blocks, tol=1e-10):
    """blocks: iterable of 9x9 complex matrices (every spin-diagonal quadrant of every hopping block).  Returns C (9x9 unitary) or None."""
    n = 9
    I = np.eye(n)
    G = np.zeros((n * n, n * n), np.complex128)
    for H in blocks:
        M = np.kron(I, H) - np.kron(np.conj(H).T, I)                               # vec(H W - W conj(H)), column-major vec
        G += M.conj().T @ M
    ev, vec = np.linalg.eigh(G)
    if ev[0] > 1e-12 * ev[-1] or ev[1] < 1e-6 * ev[-1]:                            # exactly one null vector (G = M^H M: squared singular values)
        return None
    W = vec[:, 0].reshape(n, n, order="F")
    W = W * np.sqrt(n / np.trace(W @ W.conj().T).real)
    if np.abs(W @ W.conj().T - I).max() > 1e-8:
        return None
    # W symmetric up to a phase: fix the phase so that W = W^T
    ph = np.angle(np.sum(W * np.conj(W.T)))
    W = W * np.exp(-0.5j * ph)
    if np.abs(W - W.T).max() > 1e-8:
        return None
    # Takagi of a symmetric unitary: Re W and Im W are commuting real symmetric matrices -> one real orthogonal O diagonalises both
    t = 0.7548776662466927
    _, O = np.linalg.eigh(W.real + t * W.imag)
    d = np.diag(O.T @ W @ O)
    if np.abs(O.T @ W @ O - np.diag(d)).max() > 1e-8:
        return None
    return O @ np.diag(np.exp(0.5j * np.angle(d)))


def transform_operator(p, C=None):
    """Problem dict (ee, lsham[, eeo, enim, hall, hallo]) in the basis U = diag(C, C): every block -> U^H B U; spin-diagonal quadrants of
    the hopping blocks made exactly real (their imaginary parts are rounding), spin-flip quadrants of hopping blocks must vanish.
    Returns (new problem dict, U) or (None, None) when no such basis exists."""
    ee = p["ee"]
    nb = int(p["nn"][:, 0].max())
    hop = [ee[9 * sp:9 * sp + 9, 9 * sp:9 * sp + 9, s, t] for t in range(ee.shape[3]) for s in range(1, nb) for sp in (0, 1)]
    for k in ("hall",):
        if p.get(k) is not None:
            hop += [p[k][9 * sp:9 * sp + 9, 9 * sp:9 * sp + 9, s, t] for t in range(p[k].shape[3]) for s in range(1, nb) for sp in (0, 1)]
    for k in ("eeo", "hallo"):
        if p.get(k) is not None:
            hop += [p[k][9 * sp:9 * sp + 9, 9 * sp:9 * sp + 9, s, t] for t in range(p[k].shape[3]) for s in range(0, nb) for sp in (0, 1)]
    if C is None:
        C = find_real_basis(hop)
    if C is None:
        return None, None
    U = np.kron(np.eye(2), C)
    q = dict(p)
    for k in ("ee", "eeo", "hall", "hallo"):
        if p.get(k) is not None:
            B = np.einsum("ab,bcst,cd->adst", U.conj().T, p[k], U)
            s0 = 1 if k in ("ee", "hall") else 0
            scale = np.abs(B).max()
            for sp in (0, 1):
                blk = B[9 * sp:9 * sp + 9, 9 * sp:9 * sp + 9, s0:nb, :]
                if np.abs(blk.imag).max() > 1e-12 * scale:
                    return None, None
                blk.imag[...] = 0.0
            for a, b in ((slice(0, 9), slice(9, 18)), (slice(9, 18), slice(0, 9))):
                if np.abs(B[a, b, s0:nb, :]).max() > 1e-12 * scale:
                    return None, None
                B[a, b, s0:nb, :] = 0.0
            q[k] = np.asfortranarray(B)
    for k in ("lsham", "enim"):
        if p.get(k) is not None:
            q[k] = np.asfortranarray(np.einsum("ab,bct,cd->adt", U.conj().T, p[k], U))
    return q, U
