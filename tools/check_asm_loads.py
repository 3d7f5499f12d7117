#!/usr/bin/env python3
"""ISA hazard check for hand-written vector-memory loads: a load issued from inline asm is invisible to the compiler's
wait-count model, so nothing stops the register allocator from spilling, copying or reusing its destination register between the load and the
hand-written `s_waitcnt vmcnt` that lands it -- the spill stores the register's OLD contents and the reload after the wait brings them back:
silently wrong bytes.  This script compiles a translation unit to gfx950 assembly with the product flags and walks EVERY kernel's control-flow
graph with the hardware's counter semantics (gfx9: every vector-memory instruction, loads and stores alike, increments vmcnt at issue and they
retire in order; `s_waitcnt vmcnt(k)` leaves at most the k youngest in flight): an instruction that names a register a load still in flight
will write is reported.  For compiler-issued loads the compiler guarantees this by construction, so on a unit without inline-asm loads the
check is a no-op that passes; every hit is a hand-written load whose wait the compiler did not know about.

usage: tools/check_asm_loads.py [--src conv_aux.hip] [--asm FILE.s] [--kernel-filter REGEX] [-v]      exit code 0 = clean"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PRODUCT_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]

VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
BR = re.compile(r"^s_c?branch\S*\s+(\.L\S+)")
MAX_STATES = 64  # distinct in-flight lists remembered per basic block (the kernels here need 1-6)


def regs_of(operand_text):
    out = set()
    for m in REG.finditer(operand_text):
        f = m.group(1)
        if m.group(2) is not None:
            out.add((f, int(m.group(2))))
        else:
            out.update((f, k) for k in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def parse_function(lines):
    """-> list of (label or None, [instructions]) in text order; an instruction is (mnemonic, operand text, source line number)"""
    blocks = [(None, [])]
    in_asm = False
    for no, ln in lines:
        if ";#ASMSTART" in ln:
            in_asm = True
        elif ";#ASMEND" in ln:
            in_asm = False
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            m = re.match(r"^(\.L\S+):", s)
            if m:
                blocks.append((m.group(1), []))
            continue
        m = re.match(r"^(\.L\S+):", s)
        if m:
            blocks.append((m.group(1), []))
            continue
        parts = s.split(None, 1)
        blocks[-1][1].append((parts[0], parts[1] if len(parts) > 1 else "", no, in_asm))
    return blocks


def step(state, ins, hits, name):
    """state: tuple of frozensets (destination registers of the vector-memory operations in flight, oldest first; stores: empty set)"""
    mn, ops, no, by_hand = ins
    if mn == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", ops)
        if m:
            k = int(m.group(1))
            return state[len(state) - k:] if k < len(state) else state
        if re.fullmatch(r"\s*(0|0x0+)\s*", ops) or not re.search(r"cnt\(", ops):
            try:  # numeric form: vmcnt = bits 3:0 and 15:14
                v = int(ops.strip(), 0)
                k = (v & 0xF) | (((v >> 14) & 3) << 4)
                return state[len(state) - k:] if k < len(state) else state
            except ValueError:
                pass
        return state
    used = regs_of(ops)
    if used and state:
        for pend in state:
            clash = pend & used
            if clash:
                hits.append((name, no, f"{mn} {ops}", sorted(clash)))
                break
    if VMEM.match(mn):
        # only loads written by hand are tracked by register: the compiler orders its own (and relies on in-order return for a second load into
        # the same register, which a register-level check would flag); everything else is counted only
        dst = frozenset()
        if by_hand and "_load" in mn and "_lds_" not in mn:
            dst = frozenset(regs_of(ops.split(",")[0]))
        state = state + (dst,)
        if len(state) > 63:  # the hardware counter saturates at 63: the issue stalls, i.e. the oldest have retired
            state = state[-63:]
    return state


def check_function(name, lines, verbose=False):
    blocks = parse_function(lines)
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    seen = [set() for _ in blocks]
    hits = []
    work = [(0, ())]
    overflow = False
    while work:
        bi, st = work.pop()
        # (destination sets of operations whose registers nobody can clash with any more stay in the list: they matter for the counting)
        if st in seen[bi]:
            continue
        if len(seen[bi]) >= MAX_STATES:
            overflow = True
            continue
        seen[bi].add(st)
        fall = True
        prev = ("", "")
        for ins in blocks[bi][1]:
            mn, ops = ins[0], ins[1]
            m = BR.match(f"{mn} {ops}")
            if m:
                tgt = index.get(m.group(1))
                if tgt is not None:
                    work.append((tgt, st))
                # `s_or_b64 exec, exec, saved` (the end of a masked region) + s_cbranch_execnz: exec is what it was in front of the region; a path
                # on which it is zero executes no vector instruction at all, so only the taken edge can carry a hazard
                restored = prev[0] == "s_or_b64" and prev[1].replace(" ", "").startswith("exec,exec,")
                if mn == "s_branch" or (mn == "s_cbranch_execnz" and restored):
                    fall = False
                    break
                prev = (mn, ops)
                continue
            if mn in ("s_endpgm", "s_setpc_b64"):
                fall = False
                break
            st = step(st, ins, hits, name)
            prev = (mn, ops)
        if fall and bi + 1 < len(blocks):
            work.append((bi + 1, st))
    uniq = {}
    for h in hits:
        uniq.setdefault((h[0], h[1]), h)
    if overflow and verbose:
        print(f"  note: {name}: more than {MAX_STATES} in-flight lists at one block, exploration truncated there")
    return list(uniq.values())


def functions(text):
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\S+):\s", text[i] + " ")
        if m:
            j = i + 1
            while j < len(text) and not text[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), [(k + 1, text[k]) for k in range(i + 1, j)]
            i = j
        else:
            i += 1


def compile_to_asm(src, hipcc, extra=()):
    td = tempfile.mkdtemp(prefix="asmchk_")
    out = os.path.join(td, os.path.basename(src) + ".s")
    subprocess.check_call([hipcc, *PRODUCT_FLAGS, *extra, "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=os.path.join(ROOT, "yolo_quantization_amd", "csrc", "conv_aux.hip"))
    ap.add_argument("--asm", default=None, help="check this assembly file instead of compiling --src")
    ap.add_argument("--kernel-filter", default=r".", help="regex on the mangled kernel names")
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    path = a.asm or compile_to_asm(a.src, a.hipcc, tuple(os.environ.get("EXTRA_HIPCC_FLAGS", "").split()))
    text = open(path).read().splitlines()
    bad = nk = 0
    for name, lines in functions(text):
        if not re.search(a.kernel_filter, name):
            continue
        nk += 1
        hits = check_function(name, lines, a.v)
        for _, no, ins, clash in hits[:8]:
            regs = ", ".join(f"{f}{n}" for f, n in clash)
            print(f"{name}: line {no}: `{ins}` touches {regs} while a load into it is still in flight")
        bad += len(hits)
    print(f"checked {nk} kernels of {os.path.basename(a.src if not a.asm else a.asm)}: {'clean' if not bad else str(bad) + ' hazards'}")
    return 1 if bad or not nk else 0


if __name__ == "__main__":
    sys.exit(main())
