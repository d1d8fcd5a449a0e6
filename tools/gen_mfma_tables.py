#!/usr/bin/env python3
"""Coefficient tables of acm_tile2m's first pass (libacm_amd/csrc/acm_mfma_tables.inc).

The first three stages of the subband cascade (reference src/decode.c:527-590, juggle / juggle_block) are linear, so over
one residue class of the columns the first G of them are ONE banded integer matrix: the 2 * 2^G outputs of a row pair (2 rows x
2^G columns i + q*cols/2^G) are sums of the 4 * 2^G stage-0 inputs of that row pair and of the one in front of it.  This script
derives the matrix by pushing unit impulses through the stages exactly as the oracle runs them, checks it on random input, and
writes, for G = 3 and G = 4,
  A<G>[v][m][k]     int8   coefficient of input k (k = 2^G*row + q, rows: two in front, then the pair itself) in output m
  KROW<G>[v][r][m]  int32  128 * sum of the coefficients of input row r   (staged low bytes are stored minus 128)
  BIAS<G>[v][f][m]  int32  what the "+1" of decode.c:561-564 (added after stage 0) has become at output m; f = 1: the two
                           rows in front do not exist (first row pair of a stream)
for the two storage conventions v of the LDS passes (v = 1: outputs of odd positions are kept negated, see StageKind in
acm_kernels.hip).  Run from the repo root:  python3 tools/gen_mfma_tables.py
"""
import os
import numpy as np

def cascade(G, x_prev, x_cur, bias_prev=None, bias_cur=None):
    U = 1 << G
    B = 2 * U
    v = np.concatenate([x_prev, x_cur]).astype(np.int64)
    n = len(v)
    for t in range(G):
        d = 1 << (G - 1 - t)
        pb = G - 1 - t
        out = np.zeros(n, dtype=np.int64)
        for p in range(n):
            u = p % B
            z0 = v[p]
            z1 = v[p - d] if p >= d else 0
            z2 = v[p - 2 * d] if p >= 2 * d else 0
            y = 2 * z1 + (-(z2 + z0) if (u >> pb) & 1 else (z2 + z0))
            if t == 0:
                b = bias_prev if p < B else bias_cur
                if b is not None:
                    y += b[u]
            out[p] = y
        v = out
    return v[B:]


def tables(G):
    U = 1 << G
    B = 2 * U
    A = np.zeros((B, 2 * B), dtype=np.int64)
    for j in range(2 * B):
        e = np.zeros(2 * B, dtype=np.int64)
        e[j] = 1
        A[:, j] = cascade(G, e[:B], e[B:])
    assert np.abs(A).max() <= 127
    assert not A[:, :2].any(), "reach of G stages: 2 (2^G - 1) positions"
    rng = np.random.default_rng(G)
    for _ in range(32):
        xp, xc = rng.integers(-32768, 32768, B), rng.integers(-32768, 32768, B)
        assert np.array_equal(A @ np.concatenate([xp, xc]), cascade(G, xp, xc))
    bpos = np.array([1 if u % (U // 2) == 0 else 0 for u in range(B)], dtype=np.int64)
    z = np.zeros(B, dtype=np.int64)
    return A, cascade(G, z, z, bpos, bpos), cascade(G, z, z, None, bpos)


def toeplitz(G):
    """The first G stages over one residue class as a block-Toeplitz operator: the 2^G outputs of ROW r are T0 x[r] + T1 x[r-1] + T2 x[r-2]
    (x[r] = the row's 2^G stage-0 inputs of the class; the stages' signs repeat with the row, their reach 2 (2^G - 1) positions stays inside
    two rows).  B[j] = what a "+1" added after stage 0 in row r - j (decode.c:561-564: at q = 0 and q = 2^G / 2 of residue class 0) has
    become in row r."""
    U = 1 << G

    def run(x, bias=None):
        v = np.array(x, dtype=np.int64)
        n = len(v)
        for t in range(G):
            d = 1 << (G - 1 - t)
            pb = G - 1 - t
            out = np.zeros(n, dtype=np.int64)
            for p in range(n):
                z0 = v[p]
                z1 = v[p - d] if p >= d else 0
                z2 = v[p - 2 * d] if p >= 2 * d else 0
                out[p] = 2 * z1 + (-(z2 + z0) if ((p % U) >> pb) & 1 else (z2 + z0))
            if t == 0 and bias is not None:
                out += bias
            v = out
        return v

    T = [np.zeros((U, U), dtype=np.int64) for _ in range(4)]
    for k in range(U):
        e = np.zeros(4 * U, dtype=np.int64)
        e[k] = 1
        y = run(e)
        for j in range(4):
            T[j][:, k] = y[j * U:(j + 1) * U]
    assert not T[3].any(), "reach of G stages: 2 (2^G - 1) positions < two rows"
    assert max(np.abs(t).max() for t in T) <= 127
    rng = np.random.default_rng(100 + G)
    for _ in range(8):
        x = rng.integers(-32768, 32768, 5 * U)
        y = run(x)
        for r in range(2, 5):
            assert np.array_equal(y[r * U:(r + 1) * U], T[0] @ x[r * U:(r + 1) * U] + T[1] @ x[(r - 1) * U:r * U] + T[2] @ x[(r - 2) * U:(r - 1) * U])
    B = []
    z = np.zeros(4 * U, dtype=np.int64)
    b = np.zeros(4 * U, dtype=np.int64)
    b[0] = b[U // 2] = 1                           # the "+1" of row 0
    y = run(z, b)
    for j in range(3):
        B.append(y[j * U:(j + 1) * U].copy())
    assert not y[3 * U:].any()
    return T[:3], B


def emit_toeplitz(lines, G):
    T, B = toeplitz(G)
    U = 1 << G
    signs = [np.array([-1 if (v and (q & 1)) else 1 for q in range(U)]) for v in range(2)]
    lines.append("/* [storage convention of the next pass][j: the input row is row r - j][output q][input k] */")
    lines.append("__device__ const int8_t ACM_TZ%d[2][3][%d][%d] = {" % (G, U, U))
    for v in range(2):
        lines.append("\t{")
        for j in range(3):
            lines.append("\t\t{")
            for q in range(U):
                lines.append("\t\t\t{ " + ", ".join("%d" % c for c in T[j][q] * signs[v][q]) + " },")
            lines.append("\t\t},")
        lines.append("\t},")
    lines.append("};")
    lines.append("/* [convention][rows in front of row r that exist: 0, 1, 2 or more][output q]: the \"+1\" of rows r, r - 1, r - 2 as it arrives in row r */")
    lines.append("__device__ const int32_t ACM_TZ%d_BIAS[2][3][%d] = {" % (G, U))
    for v in range(2):
        lines.append("\t{")
        acc = np.zeros(U, dtype=np.int64)
        for j in range(3):
            acc = acc + B[j]
            lines.append("\t\t{ " + ", ".join("%d" % c for c in acc * signs[v]) + " },")
        lines.append("\t},")
    lines.append("};")
    print("Toeplitz G = %d: max |coef| %d, row abs sum %d" % (G, int(max(np.abs(t).max() for t in T)), int(sum(np.abs(t) for t in T).sum(axis=1).max())))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    tz = ["/* generated by tools/gen_mfma_tables.py - do not edit */"]
    emit_toeplitz(tz, 6)
    out6 = os.path.join(here, "..", "libacm_amd", "csrc", "acm_toeplitz_tables.inc")
    with open(out6, "w") as f:
        f.write("\n".join(tz) + "\n")
    print("wrote", os.path.normpath(out6))
    lines = ["/* generated by tools/gen_mfma_tables.py - do not edit */"]
    for G in (3, 4):
        A, bias_full, bias_fresh = tables(G)
        U = 1 << G
        B = 2 * U
        signs = [np.array([-1 if (v and (m & 1)) else 1 for m in range(B)]) for v in range(2)]
        lines.append("__device__ const int8_t ACM_MF_A%d[2][%d][%d] = {" % (G, B, 2 * B))
        for v in range(2):
            Av = A * signs[v][:, None]
            lines.append("\t{")
            for m in range(B):
                lines.append("\t\t{ " + ", ".join("%d" % c for c in Av[m]) + " },")
            lines.append("\t},")
        lines.append("};")
        lines.append("__device__ const int32_t ACM_MF_KROW%d[2][4][%d] = {" % (G, B))
        for v in range(2):
            lines.append("\t{")
            for r in range(4):
                k = 128 * (A[:, U * r:U * r + U].sum(axis=1)) * signs[v]
                lines.append("\t\t{ " + ", ".join("%d" % c for c in k) + " },")
            lines.append("\t},")
        lines.append("};")
        lines.append("__device__ const int32_t ACM_MF_BIAS%d[2][2][%d] = {" % (G, B))
        for v in range(2):
            lines.append("\t{")
            for resp in (bias_full, bias_fresh):
                lines.append("\t\t{ " + ", ".join("%d" % c for c in resp * signs[v]) + " },")
            lines.append("\t},")
        lines.append("};")
        print("G = %d: max |coef| %d, row abs sum %d" % (G, int(np.abs(A).max()), int(np.abs(A).sum(axis=1).max())))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libacm_amd", "csrc", "acm_mfma_tables.inc")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
