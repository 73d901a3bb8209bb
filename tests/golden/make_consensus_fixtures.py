#!/usr/bin/env python3
"""Convert the reference's consensus-spec YAML vectors into compact fixtures.

Source (read-only, only present in the build container):
    /root/reference/tests/<api>/kzg-mainnet/<case>/data.yaml   (311 files, 80 MB)
driven in the reference by consensus_specs_test.go:19-519.

Output (committed, travels to the GPU box):
    tests/golden/consensus_vectors.json   one record per case, small fields inline (hex)
    tests/golden/consensus_pool.bin       de-duplicated pool of the big byte strings
                                          (blobs, cells), addressed by {"ref": [off, len]}

The same 7 valid + few invalid blobs recur across most APIs, so 80 MB of YAML hex
collapses to a few MB of raw bytes.  This is data only: inputs and expected outputs.

Also copies the trusted setup JSON (data file used verbatim by every vector).

Run:  python tests/golden/make_consensus_fixtures.py
"""
import glob
import hashlib
import json
import os
import shutil
import sys

import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
INLINE_MAX = 96  # bytes; anything longer goes to the pool

APIS = [
    "blob_to_kzg_commitment",
    "compute_kzg_proof",
    "compute_blob_kzg_proof",
    "verify_kzg_proof",
    "verify_blob_kzg_proof",
    "verify_blob_kzg_proof_batch",
    "compute_cells_and_kzg_proofs",
    "verify_cell_kzg_proof_batch",
    "recover_cells_and_kzg_proofs",
]

pool = bytearray()
pool_index = {}


def intern(raw: bytes):
    h = hashlib.sha256(raw).digest()
    if h not in pool_index:
        pool_index[h] = (len(pool), len(raw))
        pool.extend(raw)
    off, ln = pool_index[h]
    return {"ref": [off, ln]}


def conv(x):
    if isinstance(x, dict):
        return {k: conv(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [conv(v) for v in x]
    if isinstance(x, str):
        if not x.startswith("0x"):
            # a few invalid-length cases still are 0x-prefixed hex; anything else is kept as is
            return {"str": x}
        hx = x[2:]
        if len(hx) % 2:
            # odd-length hex cannot be bytes: keep as a string (the API must reject it;
            # at the byte-array boundary such an input is unrepresentable -> length error)
            return {"badhex": x}
        raw = bytes.fromhex(hx)
        if len(raw) > INLINE_MAX:
            return intern(raw)
        return {"hex": hx}
    return x  # bool / int / None


def main():
    out = {}
    n = 0
    for api in APIS:
        cases = []
        files = sorted(glob.glob(f"{REF}/tests/{api}/*/*/data.y*ml"))
        for f in files:
            with open(f) as fh:
                d = yaml.load(fh, Loader=yaml.CSafeLoader)
            name = os.path.basename(os.path.dirname(f))
            cases.append({"name": name, "input": conv(d["input"]), "output": conv(d["output"])})
            n += 1
        out[api] = cases
        print(f"{api}: {len(cases)} cases", file=sys.stderr)
    with open(os.path.join(HERE, "consensus_vectors.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    with open(os.path.join(HERE, "consensus_pool.bin"), "wb") as fh:
        fh.write(pool)
    print(f"{n} cases, pool {len(pool)} bytes, {len(pool_index)} unique blobs/cells", file=sys.stderr)
    # one vector as raw bytes for the plain-C caller test (tests/c_abi/use_all.c): blob | commitment | blob proof of the first valid
    # blob_to_kzg_commitment case that compute_blob_kzg_proof covers too
    def raw(x):
        return bytes.fromhex(x[2:]) if isinstance(x, str) and x.startswith("0x") and len(x) % 2 == 0 else None
    done = False
    for f in sorted(glob.glob(f"{REF}/tests/blob_to_kzg_commitment/*/*/data.yaml")):
        with open(f) as fh:
            d = yaml.load(fh, Loader=yaml.CSafeLoader)
        blob, cm = raw(d["input"]["blob"]), raw(d["output"]) if d["output"] else None
        if not blob or not cm or len(blob) != 131072 or cm[0] == 0xC0:
            continue
        for g in sorted(glob.glob(f"{REF}/tests/compute_blob_kzg_proof/*/*/data.yaml")):
            with open(g) as fh:
                e = yaml.load(fh, Loader=yaml.CSafeLoader)
            if e["output"] and raw(e["input"]["blob"]) == blob and raw(e["input"]["commitment"]) == cm:
                with open(os.path.join(HERE, "c_caller_vector.bin"), "wb") as fh:
                    fh.write(blob + cm + raw(e["output"]))
                done = True
                break
        if done:
            break
    assert done
    # the trusted setup (data): used by the oracle, the product and every vector -- ONE copy, next to the product library
    shutil.copyfile(f"{REF}/trusted_setup.json", os.path.join(os.path.dirname(os.path.dirname(HERE)), "go-eth-kzg_amd", "trusted_setup.json"))


if __name__ == "__main__":
    main()
