#!/usr/bin/env python3
"""Mint tests/golden/sha256_kat.json: known answers for the caller-side step RSASignatureVerifier::verify_pkcs1v15_signature
(reference src/lib.rs:205-239): SHA-256 of the message, digest bytes reversed and packed into four 64-bit limbs.

* the example digests of FIPS 180-4 / the NIST example-values document ("abc", the empty string, the 448-bit and 896-bit
  messages, one million 'a') -- public known answers, written out literally below and asserted against both hashlib and
  the restatement oracle/pyref.py;
* the reference's own vector: src/chip.rs:713 / 758 / 803 give the hashed message of its three RSA known-answer tests as a
  decimal integer, which is SHA-256("hello world") read big-endian -- that pins the byte order of the limb packing.

Run from the repo root:  python tests/golden/make_sha256_kat.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as R  # noqa: E402

FIPS = [
    ("abc", 1, "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
    ("", 1, "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
    ("abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", 1, "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
    ("abcdefghbcdefghicdefghijdefghijkefghijklfghijklmghijklmnhijklmnoijklmnopjklmnopqklmnopqrlmnopqrsmnopqrstnopqrstu", 1,
     "cf5b16a778af8380036ce59e7b0492370b249b11e8f07a51afac45037afee9d1"),
    ("a", 1000000, "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"),
]
# reference src/chip.rs:713 (= :758, :803): hashed_msg of test_rsa_signature_circuit1/2 and the Bad twin
REF_MSG = "hello world"
REF_HASHED = 83814198383102558219731078260892729932246618004265700685467928187377105751529


def main():
    out = {"source": "FIPS 180-4 example digests; reference src/chip.rs:713 (SHA-256 of 'hello world' as an integer)", "vectors": []}
    for text, repeat, digest_hex in FIPS + [(REF_MSG, 1, "%064x" % REF_HASHED)]:
        msg = (text * repeat).encode()
        assert hashlib.sha256(msg).hexdigest() == digest_hex, text[:16]
        assert R.sha256(msg).hex() == digest_hex, text[:16]
        st = R.Stream()
        limbs = R.hashed_msg(bytes.fromhex(digest_hex), st)
        assert sum(v << (64 * i) for i, v in enumerate(limbs)) == int(digest_hex, 16)   # src/chip.rs:141-144 reads them little-endian
        out["vectors"].append({"msg": text, "repeat": repeat, "digest": digest_hex, "hashed_limbs": [str(v) for v in limbs],
                               "stream_sha256": hashlib.sha256(bytes(st.bytes())).hexdigest()})
    assert out["vectors"][-1]["hashed_limbs"] == [str((REF_HASHED >> (64 * i)) & (2**64 - 1)) for i in range(4)]
    with open(os.path.join(HERE, "sha256_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote sha256_kat.json:", len(out["vectors"]), "vectors")


if __name__ == "__main__":
    main()
