"""TEST HELPER: the 5-column advice image of one mul_mod, built in plain Python from the ORACLE's flat stream and the
operands, following ONLY the row table documented in DESIGN.md section 2b / include/h2r.h (an independent restatement of
the placement; the values come from the oracle).  Cells are canonical field elements (32 bytes little-endian)."""
import numpy as np


def advice_image_from_stream(p, a, b, n, stream, field_modulus):
    """p: oracle params (ctypes struct with w, L, LB, WB, CB, carry_bits, carry_sub_bits, carry_nsub ...);
    a, b, n: limb lists; stream: bytes of the mul_mod flat stream.  Returns uint8 [rows, 160]."""
    w, L = p.w, p.L
    LB, WB, CB = p.LB, p.WB, p.CB
    C = 2 * L - 1
    st = bytes(stream)
    pos = 0

    def take(nb, signed=False):
        nonlocal pos
        v = int.from_bytes(st[pos:pos + nb], "little", signed=signed)
        pos += nb
        return v

    rows = []

    def row(*cells):
        cells = list(cells) + [0] * (5 - len(cells))
        rows.append([c % field_modulus for c in cells])

    def range_rows(value, subs, sub_bits):
        run = 0
        for r0 in range(0, len(subs), 4):
            chunk = subs[r0:r0 + 4]
            for k, sv in enumerate(chunk):
                run += sv << ((r0 + k) * sub_bits)
            row(*(chunk + [0] * (4 - len(chunk))), run)
        assert run == value

    q, r = [], []
    for which in (q, r):                                    # T1 / T2
        for _ in range(L):
            v = take(LB)
            subs = [take(1) for _ in range(8)]
            which.append(v)
            range_rows(v, subs, w // 8)
    ab, qn = [], []
    for (x, y, cols) in ((a, b, ab), (q, n, qn)):           # T3 / T4: column i ascending, j ascending
        for i in range(C):
            prev = 0
            for j in range(max(0, i - L + 1), min(i, L - 1) + 1):
                acc = take(WB)
                row(x[j], y[i - j], prev, acc)
                prev = acc
            cols.append(prev)
    eqb = []
    for i in range(L):                                       # T5
        v = take(WB)
        row(qn[i], r[i], v)
        eqb.append(v)
    eqb += qn[L:]
    B = 1 << w
    carry_prev, x_prev, eq_prev = 0, 0, 1
    for i in range(C):                                       # T6
        a_b = take(WB, signed=True)
        s = take(WB)
        cy = take(CB)
        c = take(LB)
        nq = take(WB)
        amnq = take(LB)
        accx = take(WB)
        qacc = take(CB)
        modacc = take(LB)
        nq2 = take(WB)
        amnq2 = take(LB)
        f1, e1 = take(1), take(1)
        row(ab[i], eqb[i], a_b)
        row(a_b, carry_prev, s)
        row(cy); row(c); row(B, cy, nq); row(s, nq, amnq); row(c, amnq)
        row(x_prev, accx)
        row(qacc); row(modacc); row(B, qacc, nq2); row(accx, nq2, amnq2); row(modacc, amnq2)
        row(c, modacc, f1)
        row(eq_prev, f1, e1)
        if i < C - 1:
            dup = take(CB)
            subs = [take(1) for _ in range(p.carry_nsub)]
            range_rows(dup, subs, p.carry_sub_bits)
            f2, e2 = take(1), take(1)
            row(cy, dup, f2)
        else:
            f2, e2 = take(1), take(1)
            row(cy, qacc, f2)
        row(e1, f2, e2)
        carry_prev, x_prev, eq_prev = cy, qacc, e2
    assert pos == len(st)
    out = np.zeros((len(rows), 160), dtype=np.uint8)
    for ri, cells in enumerate(rows):
        out[ri] = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in cells), dtype=np.uint8)
    return out
