"""TEST HELPER: place the values of an oracle flat stream into a record using ONLY the index maps
documented in include/h2r.h (independent restatement of the plane layout, in pure Python)."""
import numpy as np


def unflatten_record(stream, lo, planes):
    """stream: bytes of one mul_mod flat stream; lo: H2RLayout; planes: list of plane names (enum order).
    Returns a zero-filled record (record_stride bytes) with every plane entry written."""
    P = {n: k for k, n in enumerate(planes)}
    L, C = lo.num_limbs, lo.num_cols
    LB, WB, CB = lo.limb_bytes, lo.wide_bytes, lo.carry_bytes
    rec = np.zeros(lo.record_stride, dtype=np.uint8)
    st = np.frombuffer(bytes(stream), dtype=np.uint8)
    pos = 0

    def take(n):
        nonlocal pos
        v = st[pos:pos + n]
        pos += n
        return v

    def put(name, byte_off, v):
        o = lo.plane_off[P[name]] + byte_off
        rec[o:o + len(v)] = v

    def put_wide(name_lo, idx, v):
        put(name_lo, idx * 16, v[:16])
        if WB > 16:
            put(name_lo[:-2] + "HI", idx * 8, v[16:])

    for which in ("Q", "R"):
        for k in range(L):
            put(which, k * LB, take(LB))
            put(which + "_SUB", k * lo.plane_elem[P[which + "_SUB"]], take(lo.limb_nsub))
    for which in ("AB", "QN"):
        for i in range(C):
            j = 0 if L >= i + 1 else i + 1 - L
            while j < L and j <= i:
                v = take(WB)
                im = i % L
                g, stp = divmod(j, lo.acc_steps_per_group)
                put(which + "_LO", g * lo.acc_lo_group_bytes + stp * lo.acc_lo_row_bytes + im * 16, v[:16])
                if WB > 16:
                    put(which + "_HI", (j >> 1) * lo.acc_hi_group_bytes + im * 16 + (j & 1) * 8, v[16:])
                j += 1
    for i in range(L):
        put_wide("EQB_LO", i, take(WB))
    for i in range(C):
        put_wide("AMB_LO", i, take(WB))
        put_wide("SUM_LO", i, take(WB))
        put("CARRY", i * CB, take(CB))
        put("CMOD", i * LB, take(LB))
        put_wide("NQ1_LO", i, take(WB))
        put("AMNQ1", i * LB, take(LB))
        put_wide("ACCX_LO", i, take(WB))
        put("QACC", i * CB, take(CB))
        put("MODACC", i * LB, take(LB))
        put_wide("NQ2_LO", i, take(WB))
        put("AMNQ2", i * LB, take(LB))
        put("FLAGS", i * 4, take(2))
        if i < C - 1:
            put("CARRY_DUP", i * CB, take(CB))
            put("CARRY_SUB", i * lo.carry_sub_stride, take(lo.carry_nsub))
        put("FLAGS", i * 4 + 2, take(2))
    assert pos == len(st) == lo.stream_bytes
    return rec
