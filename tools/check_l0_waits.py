#!/usr/bin/env python3
"""Guard against a code-generation accident the first-layer kernel has met twice: the compiler's wait-count model decides that a
memory operation may still be in flight when an MFMA chain of the tile loop overwrites a register, and puts `s_waitcnt vmcnt(0)` -- a full
memory round trip, the tile's prefetch included -- INSIDE the chain (between the B-fragment reads and the chain's last MFMA).  No executed path
needs it; whether it appears depends on register allocation.  This script compiles conv_aux.hip to assembly (gfx950, the product flags), walks
every MFMA chain of conv_first_mfma_pool_kernel's instantiations and fails when a vmcnt wait sits inside one.

usage: tools/check_l0_waits.py [--kernel-filter REGEX]     exit code 0 = clean"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def chains(lines):
    """yield (first, last) line indices of runs of v_mfma separated by at most 3 non-MFMA lines"""
    idx = [i for i, ln in enumerate(lines) if "v_mfma_" in ln]
    if not idx:
        return
    start = prev = idx[0]
    for i in idx[1:]:
        if i - prev > 4:
            yield start, prev
            start = i
        prev = i
    yield start, prev


def main():
    ap = argparse.ArgumentParser()
    # (the 32-filter NM = 2 instantiations of the POOLED kernel spill at 128 registers -- scratch reloads wait on vmcnt by construction; no net
    # of BASELINE.json runs them: the 16-filter ones are what this guards)
    ap.add_argument("--kernel-filter", default=r"conv_first_mfma_pool_kernelILi\dELb[01]ELi1E", help="regex on the mangled kernel names to check")
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    a = ap.parse_args()
    src = os.path.join(ROOT, "yolo_quantization_amd", "csrc", "conv_aux.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "conv_aux.s")
        subprocess.check_call([a.hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                               "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    bad = 0
    nk = nchains = 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\S+):\s", text[i] + " ")
        if m and re.search(a.kernel_filter, m.group(1)):
            name = m.group(1)
            j = i + 1
            while j < len(text) and not text[j].startswith(".Lfunc_end"):
                j += 1
            body = [ln for ln in text[i:j] if not ln.lstrip().startswith(";")]
            nk += 1
            for first, last in chains(body):
                if last - first < 7:  # the tile loop's chains are 8 or 12 MFMAs
                    continue
                nchains += 1
                # from the B-fragment reads in front of the chain (at most 8 lines up) to its last MFMA
                lo = first
                for k in range(first - 1, max(first - 9, 0), -1):
                    if "ds_read" in body[k]:
                        lo = k
                if any("global_load" in body[k] or "scratch_load" in body[k] for k in range(lo, last)):
                    continue  # the cold exact path of a tile (it fetches its FP64 multipliers again between the MFMAs): its waits are its own
                for k in range(lo, last):
                    if re.search(r"s_waitcnt\s+vmcnt", body[k]):
                        print(f"{name}: `{body[k].strip()}` inside the MFMA chain at function line {first}..{last}")
                        bad += 1
            i = j
        else:
            i += 1
    print(f"checked {nchains} MFMA chains of {nk} kernel instantiations: {'clean' if not bad else str(bad) + ' vmcnt waits inside chains'}")
    return 1 if bad or not nchains else 0


if __name__ == "__main__":
    sys.exit(main())
