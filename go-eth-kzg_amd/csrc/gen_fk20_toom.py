#!/usr/bin/env python3
"""Generate csrc/fk20_toom.h: constants of the one-multiplication-deep G1 transform stage of FK20 (fk20_kernels.hip).

What the stage computes (reference internal/kzg_multi/fk20/fk20.go:76-90 + toeplitz.go:95-125): from the 128 MSM results u_k,
    h      = IFFT_128(u)[0..64)            (G1 inverse FFT, 1/128 included, upper half dropped)
    P      = FFT_128(h | 0 ... 0)          (G1 FFT of the zero-padded vector)
    proofs = bitrev(P).
As ONE linear map P = M u the matrix is  M[j][k] = (1/128) sum_{m<64} w^((j-k) m)  (w = w_128):
    M[j][j] = 1/2,   M[j][k] = 0 for even j - k != 0,   M[j][k] = (1/64) / (1 - w^(j-k)) for odd j - k,
i.e. P = u/2 + c (*) u, a cyclic convolution with a FIXED kernel c that couples even outputs to odd inputs and vice versa: two
64 x 64 circulant (hence Toeplitz) matrix-vector products with fixed scalar matrices and variable G1 points.

A radix-2 G1 FFT pays one 255-bit scalar multiplication per butterfly and is 14 multiplications DEEP (IFFT + FFT); on the GPU depth
is what costs (a scalar multiplication is a ~1.9 ms dependent chain for a wavefront, whatever the batch size).  A Toeplitz product
with a fixed matrix has bilinear algorithms whose pre- and post-processing use SMALL INTEGER coefficients only: Toom-Cook in its
transposed ("middle product") form.  For a 4-way split with the evaluation points (x : z) below,
    y_{3-i} = sum_r  e_r[i] * ( T~_r  (sum_j e_r[j] u_j) ),    e_r[i] = x_r^i z_r^(3-i),    T~_r = sum_k G[k][r] T_{3-k}
where T_d are the seven Toeplitz blocks of the matrix and G the 7 x 7 interpolation matrix -- rational, but it only ever multiplies
the FIXED side.  Three levels (64 = 4^3) leave 7^3 = 343 products of one fixed scalar with one point per block: 686 scalar
multiplications per blob (the radix-2 FFTs need 642 non-trivial ones), all independent -- ONE multiplication deep -- framed by
additions and doublings of points (coefficients 1, 2, 4, 8 and signs).

This script derives the 686 scalars S, checks the whole construction against the definition on random inputs in Fr (everything is
linear, so scalars can stand in for points), and writes
    FK20_TOOM_WNAF[686][66] : the scalars' GLV halves as width-5 NAF digit strings (layout of G1TW_WNAF_FWD in constants29.h)
    FK20_PRE_PROG / FK20_POST_PROG : the small programs of the addition networks (shared by every level).
"""
import os
import random

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ROOT = 10238227357739495823651030575849232062558860180284477541189508159991286009131
X2 = 0xD201000000010000 ** 2

# evaluation points (x : z) of the 4-way Toom split, in the order the kernels index them
POINTS = [(0, 1), (1, 0), (1, 1), (-1, 1), (2, 1), (-2, 1), (1, 2)]
NR = len(POINTS)


def inv(a):
    return pow(a % R, -1, R)


def e_vec(pt, deg):
    x, z = pt
    return [(x ** i) * (z ** (deg - i)) for i in range(deg + 1)]


E3 = [e_vec(p, 3) for p in POINTS]          # 7 x 4: evaluation of a degree-3 "polynomial" of blocks
E6 = [e_vec(p, 6) for p in POINTS]          # 7 x 7: evaluation of the degree-6 product


def mat_inv_mod(m):
    n = len(m)
    a = [[v % R for v in row] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(m)]
    for col in range(n):
        piv = next(r for r in range(col, n) if a[r][col])
        a[col], a[piv] = a[piv], a[col]
        iv = inv(a[col][col])
        a[col] = [v * iv % R for v in a[col]]
        for r in range(n):
            if r != col and a[r][col]:
                f = a[r][col]
                a[r] = [(v - f * w) % R for v, w in zip(a[r], a[col])]
    return [row[n:] for row in a]


# C_k = sum_r G[k][r] * C(point r): G = (E6 as a matrix [r][k])^-1, i.e. G[k][r]
G = mat_inv_mod(E6)


def toom_scalars(t, n):
    """leaves of the recursion for the Toeplitz matrix with entries t[d], d in (-n, n): list of 7^log4(n) scalars in path order"""
    if n == 1:
        return [t[0] % R]
    m = n // 4
    out = []
    for r in range(NR):
        sub = {dd: sum(G[k][r] * t[(3 - k) * m + dd] for k in range(7)) % R for dd in range(-(m - 1), m)}
        out += toom_scalars(sub, m)
    return out


def toom_apply(S, u, n):
    """y = T u through pre-network, pointwise products with the leaves S, post-network (the structure the kernels run)"""
    if n == 1:
        return [S[0] * u[0] % R]
    m = n // 4
    per = len(S) // NR
    y = [0] * n
    for r in range(NR):
        eu = [sum(E3[r][j] * u[j * m + a] for j in range(4)) % R for a in range(m)]
        z = toom_apply(S[r * per:(r + 1) * per], eu, m)
        for i in range(4):
            if E3[r][i]:
                for a in range(m):
                    y[(3 - i) * m + a] = (y[(3 - i) * m + a] + E3[r][i] * z[a]) % R
    return y


def kernels():
    w = pow(ROOT, (1 << 32) // 128, R)
    C = {d: inv(32) * inv(1 - pow(w, d, R)) % R for d in range(1, 128, 2)}          # 2 c_d = (1/32) / (1 - w^d), odd d
    gE = [C[(2 * e - 1) % 128] for e in range(64)]          # even outputs from odd inputs
    gO = [C[(2 * e + 1) % 128] for e in range(64)]          # odd outputs from even inputs
    return w, gE, gO


def reference_map(w, u):
    """fk20.go:76-90 / toeplitz.go:110-125 on scalars: IFFT_128, keep 64, FFT_128 (natural order, before the final BitReverse)"""
    n_inv = inv(128)
    h = [n_inv * sum(pow(w, (-m * k) % 128, R) * u[k] for k in range(128)) % R for m in range(64)]
    return [sum(pow(w, (j * m) % 128, R) * h[m] for m in range(64)) % R for j in range(128)]


def wnaf5(k):
    ds = []
    while k:
        if k & 1:
            d = k & 31
            if d >= 16:
                d -= 32
            k -= d
        else:
            d = 0
        ds.append(d)
        k >>= 1
    return ds


def check_incomplete_chain(k, d1, d2):
    """g1.hpp mul_wnaf_coz adds with INCOMPLETE formulas: for a point P of order r the chain acc = 2 acc; acc += d1 P; acc += d2 [x^2] P
    must never add an operand equal to +-acc, nor add to the identity after the first addition.  k is a constant, so this is decided
    here, once, for every one of the 686 chains (acc = [m] P with m known modulo r)."""
    m, started = 0, False
    for i in range(max(len(d1), len(d2)) - 1, -1, -1):
        if started:
            m = 2 * m % R
        for d, unit in ((d1[i] if i < len(d1) else 0, 1), (d2[i] if i < len(d2) else 0, X2)):
            if d == 0:
                continue
            t = d * unit % R
            if started:
                assert m % R != 0 and m != t and m != (R - t) % R, ("exceptional addition in the chain of", hex(k), i)
                m = (m + t) % R
            else:
                m, started = t, True
    assert m == k % R


def wnaf_row(k):
    d1, d2 = wnaf5(k % X2), wnaf5(k // X2)
    assert k // X2 < (1 << 128) and len(d1) <= 131 and len(d2) <= 131
    check_incomplete_chain(k, d1, d2)
    d1 += [0] * (132 - len(d1))
    d2 += [0] * (132 - len(d2))
    codes = [(a & 0xFF) | ((b & 0xFF) << 8) for a, b in zip(d1, d2)]
    return "{" + ", ".join("0x%08xu" % (codes[2 * i] | codes[2 * i + 1] << 16) for i in range(66)) + "}"


# ---- programs of the addition networks: a list of (op, operand) per output kind, interpreted by k_fk20_pre / k_fk20_post
OP_LOAD, OP_ADD, OP_SUB, OP_DBL, OP_NEG, OP_END = 0, 1, 2, 3, 4, 7


def pre_programs():
    """E_r(u) = sum_j e_r[j] u_j for the seven points, by Horner (operands: j = 0..3)"""
    progs = []
    for (x, z) in POINTS:
        if (x, z) == (0, 1):
            p = [(OP_LOAD, 0)]
        elif (x, z) == (1, 0):
            p = [(OP_LOAD, 3)]
        elif z == 1:                         # ((u3 x + u2) x + u1) x + u0
            p = [(OP_LOAD, 3)]
            for j in (2, 1, 0):
                if abs(x) == 2:
                    p.append((OP_DBL, 0))
                if x < 0:
                    p.append((OP_NEG, 0))
                p.append((OP_ADD, j))
        else:                                # (1 : 2): ((u0 2 + u1) 2 + u2) 2 + u3
            assert (x, z) == (1, 2)
            p = [(OP_LOAD, 0)]
            for j in (1, 2, 3):
                p += [(OP_DBL, 0), (OP_ADD, j)]
        progs.append(p)
    return progs


def post_programs():
    """y_A = sum_r e_r[3 - A] Z_r for A = 0..3, by Horner in powers of two (operands: r = 0..6)"""
    progs = []
    for A in range(4):
        i = 3 - A
        terms = {}
        for r in range(NR):
            cf = E3[r][i]
            if cf:
                k = abs(cf).bit_length() - 1
                assert abs(cf) == 1 << k
                terms.setdefault(k, []).append((r, cf < 0))
        p, first = [], True
        for k in range(max(terms), -1, -1):
            if not first:
                p.append((OP_DBL, 0))
            for (r, negative) in terms.get(k, []):
                if first:
                    p.append((OP_LOAD, r))
                    if negative:
                        p.append((OP_NEG, 0))
                    first = False
                else:
                    p.append((OP_SUB if negative else OP_ADD, r))
        progs.append(p)
    return progs


def run_prog(prog, operands):
    acc = 0
    for op, x in prog:
        if op == OP_LOAD:
            acc = operands[x]
        elif op == OP_ADD:
            acc = (acc + operands[x]) % R
        elif op == OP_SUB:
            acc = (acc - operands[x]) % R
        elif op == OP_DBL:
            acc = 2 * acc % R
        elif op == OP_NEG:
            acc = -acc % R
    return acc


def device_model(SE, SO, u2):
    """the exact data flow of the kernels (axis transforms with the programs, layouts as in fk20_kernels.hip) on scalars; u2 = u / 2"""
    PRE, POST = pre_programs(), post_programs()

    def pre_axis(src, npre, npost):
        dst = [0] * (npre * 7 * npost)
        for p in range(npre):
            for r in range(7):
                for q in range(npost):
                    dst[(p * 7 + r) * npost + q] = run_prog(PRE[r], [src[(p * 4 + j) * npost + q] for j in range(4)])
        return dst

    def post_axis(src, npre, npost):
        dst = [0] * (npre * 4 * npost)
        for p in range(npre):
            for A in range(4):
                for q in range(npost):
                    dst[(p * 4 + A) * npost + q] = run_prog(POST[A], [src[(p * 7 + r) * npost + q] for r in range(7)])
        return dst

    P = [0] * 128
    for blk, S in ((0, SE), (1, SO)):
        x = [u2[2 * b + (1 - blk)] for b in range(64)]          # block 0 (even outputs) reads the odd inputs and vice versa
        x = pre_axis(x, 1, 16)
        x = pre_axis(x, 7, 4)
        x = pre_axis(x, 49, 1)
        z = [S[i] * x[i] % R for i in range(343)]
        z = post_axis(z, 49, 1)
        z = post_axis(z, 7, 4)
        z = post_axis(z, 1, 16)
        for a in range(64):
            P[2 * a + blk] = (u2[2 * a + blk] + z[a]) % R
    return P


def main():
    w, gE, gO = kernels()
    tE = {d: gE[d % 64] for d in range(-63, 64)}
    tO = {d: gO[d % 64] for d in range(-63, 64)}
    SE, SO = toom_scalars(tE, 64), toom_scalars(tO, 64)
    assert len(SE) == 343 and len(SO) == 343
    rnd = random.Random(20)
    for trial in range(3):
        u = [rnd.randrange(R) for _ in range(128)] if trial else [1] + [0] * 127
        ref = reference_map(w, u)
        u2 = [v * inv(2) % R for v in u]
        # (a) the recursion itself
        xo, xe = [u2[2 * b + 1] for b in range(64)], [u2[2 * b] for b in range(64)]
        yE, yO = toom_apply(SE, xo, 64), toom_apply(SO, xe, 64)
        assert all((u2[2 * a] + yE[a]) % R == ref[2 * a] and (u2[2 * a + 1] + yO[a]) % R == ref[2 * a + 1] for a in range(64))
        # (b) the flattened data flow the kernels execute
        assert device_model(SE, SO, u2) == ref
    pre, post = pre_programs(), post_programs()
    maxlen = max(len(p) for p in pre + post) + 1

    def prog_rows(progs):
        rows = []
        for p in progs:
            codes = [(op | (x << 4)) for op, x in p] + [OP_END] * (maxlen - len(p))
            rows.append("{" + ", ".join(str(c) for c in codes) + "}")
        return "{" + ", ".join(rows) + "}"

    out = ["/* GENERATED by go-eth-kzg_amd/csrc/gen_fk20_toom.py -- do not edit.  See that script for the derivation and its self-check. */\n",
           "#pragma once\n#include <stdint.h>\n\nnamespace kzg {\n\n",
           "constexpr int FK20_TOOM_NR = 7;            /* evaluation points of a 4-way split: (0:1) (1:0) (1:1) (-1:1) (2:1) (-2:1) (1:2) */\n",
           "constexpr int FK20_TOOM_TERMS = 343;       /* 7^3 products per 64 x 64 block; two blocks per blob */\n",
           f"constexpr int FK20_PROG_LEN = {maxlen};\n",
           "/* addition-network programs: op | operand << 4; ops 0 LOAD, 1 ADD, 2 SUB, 3 DBL, 4 NEG, 7 END */\n",
           f"__device__ const uint8_t FK20_PRE_PROG[7][{maxlen}] = {prog_rows(pre)};\n",
           f"__device__ const uint8_t FK20_POST_PROG[4][{maxlen}] = {prog_rows(post)};\n",
           "/* width-5 NAF digit strings of the GLV halves of the 686 fixed scalars (block 0: even outputs from odd inputs, block 1: odd from even);\n"
           "   bit position i of row t: d1 | d2 << 8 (two int8), two positions per 32-bit word */\n",
           "__device__ const uint32_t FK20_TOOM_WNAF[686][66] = {\n    " + ",\n    ".join(wnaf_row(s) for s in SE + SO) + "};\n",
           "\n}  // namespace kzg\n"]
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "fk20_toom.h"), "w") as fh:
        fh.write("".join(out))
    ops_pre = sum(len(p) - 1 for p in pre)
    ops_post = sum(len(p) - 1 for p in post)
    print(f"fk20_toom.h written: 686 scalars, programs <= {maxlen - 1} ops (pre {ops_pre} / post {ops_post} ops per element-node), self-check passed")


if __name__ == "__main__":
    main()
