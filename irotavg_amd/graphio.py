"""Text formats of the reference's `l1_irls` demo (ral/test.cpp): input graph
(:161-228) and output rotations + weights (:314-326). Host-side plumbing.

File order is `w x y z`; in-memory order is [x y z w] (ral/test.cpp:188-193,219-221,317-321).
"""
import numpy as np


class GraphFileError(ValueError):
    """The reference prints to stderr and calls exit(-1) in these cases."""


def read_ravg_input(path):
    """Parse `m n f` / m x `i j w x y z` / up to n x `w x y z`.

    Vertex ids are arbitrary ints remapped to their sorted rank (ral/test.cpp:202-213).
    Returns dict(m, n, f, I (m,2) int32, QQ (m,4), Q (n,4) zeros where not given, n_abs_read).
    """
    with open(path, "r") as fh:
        tok = fh.read().split()
    if len(tok) < 3:
        raise GraphFileError("Corrupt input file: missing header")
    m, n, f = int(tok[0]), int(tok[1]), int(tok[2])
    pos = 3
    if len(tok) < pos + 6 * m:
        raise GraphFileError("Corrupt input file: inconsistent number of connections.")
    body = np.array(tok[pos:pos + 6 * m], dtype=np.float64).reshape(m, 6)
    pos += 6 * m
    ids = body[:, :2].astype(np.int64)
    QQ = body[:, [3, 4, 5, 2]].copy()
    verts = np.unique(ids)
    I = np.searchsorted(verts, ids).astype(np.int32)
    rest = tok[pos:]
    n_abs = min(n, len(rest) // 4)
    Q = np.zeros((n, 4))
    if n_abs > 0:
        a = np.array(rest[:4 * n_abs], dtype=np.float64).reshape(n_abs, 4)
        Q[:n_abs] = a[:, [1, 2, 3, 0]]
    if n_abs < f:
        raise GraphFileError("Insuficient number of absolute rotations. At least %d must be given." % f)
    if n != int(I[:, 1].max()) + 1:  # ral/test.cpp:236-247
        raise GraphFileError("Corrupt input file: check abs rotations")
    return dict(m=m, n=n, f=f, I=I, QQ=QQ, Q=Q, n_abs_read=n_abs)


def write_ravg_input(path, I, QQ, Q_abs, n, f):
    with open(path, "w") as fh:
        fh.write("%d %d %d\n" % (len(I), n, f))
        for (i, j), q in zip(I, QQ):
            fh.write("%d %d %.17g %.17g %.17g %.17g\n" % (i, j, q[3], q[0], q[1], q[2]))
        for q in Q_abs:
            fh.write("%.17g %.17g %.17g %.17g\n" % (q[3], q[0], q[1], q[2]))


def write_l1_irls_out(path, Q, weights):
    """n rows `w x y z` then m weights, full precision (ral/test.cpp:314-326)."""
    with open(path, "w") as fh:
        for q in Q:
            fh.write("%.17g %.17g %.17g %.17g\n" % (q[3], q[0], q[1], q[2]))
        for w in weights:
            fh.write("%.17g\n" % w)


def read_l1_irls_out(path, n):
    a = np.loadtxt(path, dtype=str, delimiter="\n") if False else None
    with open(path) as fh:
        tok = fh.read().split()
    Q = np.array(tok[:4 * n], dtype=np.float64).reshape(n, 4)[:, [1, 2, 3, 0]]
    w = np.array(tok[4 * n:], dtype=np.float64)
    return Q, w
