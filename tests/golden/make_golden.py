#!/usr/bin/env python3
"""Mint tests/golden/*.json.

Inputs are the known-answer DATA held by the reference's own tests (numbers only, cited by
reference file:line); expected outputs are derived with the independent Python big-int restatement
oracle/pyref.py and, where the reference states the answer itself (un-carried product columns,
`is_valid`, mul_mod identities, parameter goldens), taken from the reference and asserted here.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as R  # noqa: E402

# --- reference src/chip.rs:703, 708, 713 (test_rsa_signature_circuit1: valid) -------------------
KAT1_N = 27333278531038650284292446400685983964543820405055158402397263907659995327446166369388984969315774410223081038389734916442552953312548988147687296936649645550823280957757266695625382122565413076484125874545818286099364801140117875853249691189224238587206753225612046406534868213180954324992542640955526040556053150097561640564120642863954208763490114707326811013163227280580130702236406906684353048490731840275232065153721031968704703853746667518350717957685569289022049487955447803273805415754478723962939325870164033644600353029240991739641247820015852898600430315191986948597672794286676575642204004244219381500407
KAT1_SIG = 27166015521685750287064830171899789431519297967327068200526003963687696216659347317736779094212876326032375924944649760206771585778103092909024744594654706678288864890801000499430246054971129440518072676833029702477408973737931913964693831642228421821166326489172152903376352031367604507095742732994611253344812562891520292463788291973539285729019102238815435155266782647328690908245946607690372534644849495733662205697837732960032720813567898672483741410294744324300408404611458008868294953357660121510817012895745326996024006347446775298357303082471522757091056219893320485806442481065207020262668955919408138704593
# --- reference src/chip.rs:748, 753, 758 (test_rsa_signature_circuit2: valid) -------------------
KAT2_N = 24226501697440012621102249466312043787685293040734225606346036389705515508545746221669035424138747582133889500686654172873671086178893587422987328751464627501601101326475761646014534358699943642495332701081302954020983110372109611581202820849485662540890985814355975252780310958088652613376767040069489530039075302709233494829280591680666351811024913107949144932224439129715181798714328219977771472462901856297952813239115577652450722815852332547886777292613005505949100406231716599634852632308325816916535875123863510650526931916871614411907700873376659841257216885666098127478325534982891697988739616416855214839339
KAT2_SIG = 18928545496959757512579438348223103860103247450097569223971486743312798156950374943336714741350742176674694049986481729075548718599712271054643150030165230392897481507710187505775911256946250999396358633095137650326818007610162375520522758780751710735664264200260854016867498935206556916247099180950775474524799944404833222133011134000549939512938205188018503377612813102061504146765520561811620128786062447005833886367575841545493555268747671930923697279690399480501746857825917608323993022396398648205737336204493624060285359455268389160802763426461171262704764369336704988874821898000892148693988241020931055723252
# --- reference src/chip.rs:798 (test_bad_rsa_signature_circuit2: invalid, one digit off KAT2) ---
BAD_SIG = 18928545496959756512579438348223103860103247450097569223971486743312798156950374943336714741350742176674694049986481729075548718599712271054643150030165230392897481507710187505775911256946250999396358633095137650326818007610162375520522758780751710735664264200260854016867498935206556916247099180950775474524799944404833222133011134000549939512938205188018503377612813102061504146765520561811620128786062447005833886367575841545493555268747671930923697279690399480501746857825917608323993022396398648205737336204493624060285359455268389160802763426461171262704764369336704988874821898000892148693988241020931055723252
# --- reference src/chip.rs:713 / 758 / 803: SHA-256("hello world") as an integer ------------------
HASHED = 83814198383102558219731078260892729932246618004265700685467928187377105751529
E_FIX = 65537  # src/chip.rs:623 DEFAULT_E

# --- reference src/big_integer/chip.rs:2932-2947 (a) and :2951-3012 (31 un-carried columns) ------
CASE5_A = [4819187580044832333, 9183764011217009606, 11426964127496009747, 17898263845095661790,
           12102522037140783322, 4029304176671511763, 11339410859987005436, 12120243430436644729,
           2888435820322958146, 7612614626488966390, 3872170484348249672, 9589147526444685354,
           16391157694429928307, 12256166884204507566, 4257963982333550934, 916988490704]
CASE5_COLS = [23224568931658367244754058218082222889, 88516562921839445888640380379840781596,
              194478888615417946406783868151393774738, 382395265476432217957523230769986571504,
              575971019676008360859069855433378813941, 670174995752918677131397897218932582682,
              780239872348808029089572423614905198300, 850410093737715640261630122959874522628,
              800314959349304909735238452892956199392, 906862855407309870283714027678210238070,
              967727310654811444144097720329196927129, 825671020037461535758117365587238596380,
              991281789723902700168027417052185830252, 1259367815833216292413970809061165585320,
              1351495628781923848799708082622582598675, 1451028634949220760698564802414695011932,
              1290756126635958771067082204577975256756, 936482288980049848345464202850902738826,
              886330568585033438612679243731110283692, 823948310509772835433730556487356331346,
              649341353489205691855914543942648985328, 497838205323760437611385487609464464168,
              430091148520710550273018448938020664564, 474098876922017329965321439330710234148,
              536697574159375092388958994084813127393, 483446024935732188792400155524449880972,
              289799562463011227421662267162524920264, 104372664369829937912234314161010649544,
              18130279752377737976455635841349605284, 7809007931264072381739139035072,
              840867892083599894415616]
# small polynomial cases: reference src/big_integer/chip.rs:2844-2863, 2887-2903, 3038-3056, 3080-3098
POLY_CASES = {
    "case3": ([1, 0, 3], [3, 1, 0], [3, 1, 9, 3]),
    "case4": ([3, 4, 5, 6], [9, 10, 11, 12], [27, 66, 118, 184, 163, 126, 72]),
    "case6": ([1, 1], [1, 1, 1], [1, 2, 2, 1]),
    "case7": ([1, 7], [1, 1, 1], [1, 8, 8, 7]),
}


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def hexl(limbs):
    return ["%x" % v for v in limbs]


def main():
    p = R.Params(64, 32)
    L, w = 32, 64
    out = {}

    # ---- RSA KATs (src/chip.rs:683-816): expected is_valid = 1, 1, 0 ---------------------------
    kats = []
    for name, n, sig, expect in (("KAT1", KAT1_N, KAT1_SIG, 1), ("KAT2", KAT2_N, KAT2_SIG, 1),
                                 ("BAD", KAT2_N, BAD_SIG, 0)):
        st = R.Stream()
        mm = []
        x, nl = R.to_limbs(sig, L, w), R.to_limbs(n, L, w)
        # replay pow_mod_fixed_exp keeping per-mul_mod digests
        acc, squared = R.to_limbs(1, L, w), list(x)
        for bit in R.fixed_exp_bits(E_FIX):
            cur = squared
            s1 = R.Stream()
            squared = R.mul_mod(p, cur, cur, nl, s1)
            mm.append({"op": "square", "r": hexl(squared), "q": hexl(R.to_limbs(R.from_limbs(cur, w) ** 2 // n, L, w)),
                       "sha256": sha(s1.bytes())})
            st.buf += s1.buf
            if bit:
                s2 = R.Stream()
                a_old = acc
                acc = R.mul_mod(p, acc, cur, nl, s2)
                mm.append({"op": "mul", "r": hexl(acc),
                           "q": hexl(R.to_limbs(R.from_limbs(a_old, w) * R.from_limbs(cur, w) // n, L, w)),
                           "sha256": sha(s2.bytes())})
                st.buf += s2.buf
        for v in acc:
            st.put(v, p.LB)
        powed = R.from_limbs(acc, w)
        assert powed == pow(sig, E_FIX, n) == R.big_pow_mod(sig, E_FIX, n)
        assert len(mm) == 19 and len(st.buf) == 19 * 64338 + 256
        em = R.Stream()
        is_valid = R.pkcs1v15_em_check(acc, R.to_limbs(HASHED, 4, w), 2048, em)
        assert is_valid == expect, name
        inf = R.Stream()
        assert R.assert_in_field(p, x, nl, inf) == 1
        kats.append({"name": name, "n": str(n), "sig": str(sig), "hashed": str(HASHED), "e": E_FIX,
                     "is_valid": expect, "powed_limbs": hexl(acc),
                     "pow_stream_bytes": len(st.buf), "pow_stream_sha256": sha(st.bytes()),
                     "mul_mods": mm,
                     "em_stream_sha256": sha(em.bytes()), "em_stream_bytes": len(em.buf),
                     "in_field_stream_sha256": sha(inf.bytes()), "in_field_stream_bytes": len(inf.buf)})
    out["rsa_kats"] = kats

    # ---- un-carried product columns (big_integer/chip.rs:2797-3107) ------------------------------
    muls = []
    cols5 = R.mul_columns(CASE5_A, CASE5_A, None, p.WB)
    assert cols5 == CASE5_COLS, "case5 columns disagree with the reference's constants"
    muls.append({"name": "case5", "a": hexl(CASE5_A), "b": hexl(CASE5_A), "cols": [str(c) for c in CASE5_COLS]})
    for name, (a, b, cols) in POLY_CASES.items():
        got = R.mul_columns(a, b, None, p.WB)
        cols = cols + [0] * (len(got) - len(cols))   # trailing zero columns of the zero-padded limbs
        assert got == cols, name
        muls.append({"name": name, "a": hexl(a), "b": hexl(b), "cols": [str(c) for c in cols]})
    muls.append({"name": "case1", "a": ["1"], "b": ["1"], "cols": ["1"]})  # :2797-2829, 1*1 = 1
    out["mul_cases"] = muls

    # ---- mul_mod identities (big_integer/chip.rs:3123, 3164, 3204, 3246) on n = KAT1_N -----------
    ids = []
    n = KAT1_N
    for name, a, b, r in (("0*b=0", 0, KAT1_SIG % n, 0), ("n*1=0", n, 1, 0),
                          ("(n-1)^2=1", n - 1, n - 1, 1), ("(n-1)(n-2)=2", n - 1, n - 2, 2)):
        st = R.Stream()
        rr = R.mul_mod(p, R.to_limbs(a, L, w), R.to_limbs(b, L, w), R.to_limbs(n, L, w), st)
        assert R.from_limbs(rr, w) == r, name
        ids.append({"name": name, "a": str(a), "b": str(b), "n": str(n), "r": str(r), "stream_sha256": sha(st.bytes())})
    out["mul_mod_identities"] = ids

    # ---- parameter goldens (big_integer/chip.rs:1220-1249; mod.rs:504-509) -------
    table = []
    for (ww, bits) in ((64, 1024), (64, 2048), (64, 4096), (32, 4096), (32, 2048)):
        LL = bits // ww
        pp = R.Params(ww, LL)
        comp, over = R.compute_range_lens(ww, LL)
        table.append({"w": ww, "bits": bits, "L": LL, "comp": comp, "over": over,
                      "word_max_bits": pp.word_max.bit_length(), "carry_bits": pp.carry_bits,
                      "LB": pp.LB, "WB": pp.WB, "CB": pp.CB, "mul_mod_stream_bytes": pp.mul_mod_stream_bytes})
    assert table[1]["comp"] == [8, 1, 8] and table[1]["over"] == [0, 0, 6] and table[1]["carry_bits"] == 70
    assert table[3]["comp"] == [4, 1, 5] and table[3]["over"] == [0, 0, 0] and table[3]["carry_bits"] == 40
    out["params"] = table
    out["rsa_range_lens_2048"] = list(R.rsa_compute_range_lens(32))
    out["refresh_aux_32_1_1"] = R.refresh_aux_increased_limbs(32, 1, 1)
    assert out["refresh_aux_32_1_1"] == [1, 0]   # big_integer/mod.rs:509

    # ---- variable-exponent pow (src/chip.rs:283, 327: 5-bit e) on KAT1 inputs ---------------------
    var = []
    for e in (0, 1, 19, 31):
        st = R.Stream()
        res = R.pow_mod_var(p, R.to_limbs(KAT1_SIG, L, w), [e], R.to_limbs(KAT1_N, L, w), 5, st)
        assert R.from_limbs(res, w) == pow(KAT1_SIG, e, KAT1_N)
        var.append({"e": e, "exp_limb_bits": 5, "result_limbs": hexl(res), "stream_bytes": len(st.buf),
                    "stream_sha256": sha(st.bytes())})
    out["pow_var_kat1"] = var

    # ---- RSA-4096 / 32-bit limbs (BASELINE config 4) on a fixed synthetic input --------------------
    p4 = R.Params(32, 128)
    n4 = (KAT1_N << 2048) | KAT2_N | 1
    x4 = (KAT2_SIG << 2040) ^ KAT1_SIG
    x4 %= n4
    st = R.Stream()
    res = R.pow_mod_fixed_exp(p4, R.to_limbs(x4, 128, 32), E_FIX, R.to_limbs(n4, 128, 32), st)
    assert R.from_limbs(res, 32) == pow(x4, E_FIX, n4)
    out["rsa4096_w32"] = {"n": str(n4), "x": str(x4), "e": E_FIX, "result": str(R.from_limbs(res, 32)),
                          "stream_bytes": len(st.buf), "stream_sha256": sha(st.bytes())}

    # plain-text limb fixture for the C++ host-mirror test (tests/cpp/test_rsa_chip.cpp)
    with open(os.path.join(HERE, "rsa_kats_limbs.txt"), "w") as f:
        f.write("# name is_valid | 32 n limbs | 32 sig limbs | 4 hashed limbs   (hex, little-endian 64-bit limbs)\n")
        for k in kats:
            nl = R.to_limbs(int(k["n"]), 32, 64); sl = R.to_limbs(int(k["sig"]), 32, 64); hl = R.to_limbs(int(k["hashed"]), 4, 64)
            f.write("%s %d %s %s %s\n" % (k["name"], k["is_valid"], " ".join("%x" % v for v in nl), " ".join("%x" % v for v in sl),
                                          " ".join("%x" % v for v in hl)))
    path = os.path.join(HERE, "halo2_rsa_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
