#!/usr/bin/env python3
"""check_kernels.py -- build-time gate on the SHAPE of every kernel of libebo_hip.so.

Reads the device assembly `make asm` writes (csrc/ebo_kernels.s: hipcc -S --cuda-device-only of the one
translation unit that holds every kernel) and fails when a kernel

  * uses a dynamic stack                  (.amdhsa_uses_dynamic_stack 1: recursion / indirect calls),
  * calls an out-of-line device function  (s_swappc_b64 in its body, or any non-kernel function symbol in
                                           the code object: the compiler outlined something that the
                                           source means to be inlined -- its LDS arrays then travel as
                                           generic pointers and every access becomes a flat_* instruction),
  * adds to memory through a flat pointer (flat_atomic_*: an atomic the compiler could not prove to be an
                                           LDS or a global one),
  * needs more scratch than the budget    (.amdhsa_private_segment_fixed_size, bytes per lane; default 128),
  * spills vector registers               ("Folded Spill" scratch stores of VGPRs; default 0 allowed).

Why: round 2's first k_solve_edge reached the edge objective from three sites of its solver loop; the
compiler kept one out-of-line copy, passed the workgroup's LDS arrays as generic pointers, spilled 340
VGPRs -- and the kernel faulted on the device (profiles/HISTORY_kernels.md).  tools/probe/solve_edge_outlined.hip
rebuilds that shape; `check_kernels.py --expect-fail` on its assembly is part of the test-suite, so the
gate is known to catch it.  Runs on the CPU box; nothing here touches a GPU.

usage: check_kernels.py FILE.s [--scratch-budget BYTES] [--vgpr-spill-budget N] [--expect-fail] [--json]
"""
import argparse
import json
import re
import sys


def parse(path):
    text = open(path).read()
    lines = text.split("\n")
    kernels = {}
    # kernel descriptors
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)

        def field(key, default=0):
            mm = re.search(r"\.amdhsa_%s\s+(\S+)" % key, body)
            return int(mm.group(1), 0) if mm else default

        kernels[name] = {
            "scratch_bytes_per_lane": field("private_segment_fixed_size"),
            "dynamic_stack": field("uses_dynamic_stack"),
            "next_free_vgpr": field("next_free_vgpr"),
            "lds_static_bytes": field("group_segment_fixed_size"),
            "calls": 0, "flat_atomics": 0, "flat_accesses": 0, "vgpr_spill_stores": 0, "vgpr_spill_loads": 0,
        }
    # function symbols of the code object: kernels and anything the compiler left out of line
    functions = re.findall(r"^\s*\.type\s+(\S+),@function", text, re.M)
    outlined = [f for f in functions if f not in kernels]
    # bodies: from the symbol's label to its .Lfunc_end
    current = None
    for ln in lines:
        mm = re.match(r"^(\S+):\s*(;.*)?$", ln)
        if mm and mm.group(1) in kernels:
            current = mm.group(1)
            continue
        if mm and mm.group(1) in outlined:
            current = None
            continue
        if ln.startswith(".Lfunc_end"):
            current = None
            continue
        if current is None:
            continue
        k = kernels[current]
        ins = ln.strip()
        if ins.startswith("s_swappc_b64"):
            k["calls"] += 1
        elif ins.startswith("flat_atomic"):
            k["flat_atomics"] += 1
            k["flat_accesses"] += 1
        elif ins.startswith("flat_"):
            k["flat_accesses"] += 1
        elif ins.startswith("scratch_store") and "Folded Spill" in ins:
            k["vgpr_spill_stores"] += 1
        elif ins.startswith("scratch_load") and "Folded Reload" in ins:
            k["vgpr_spill_loads"] += 1
    return kernels, outlined


def violations(kernels, outlined, scratch_budget, spill_budget):
    out = []
    for f in outlined:
        out.append(("<code object>", "out-of-line device function %s" % f))
    for name, k in sorted(kernels.items()):
        if k["dynamic_stack"]:
            out.append((name, "dynamic stack"))
        if k["calls"]:
            out.append((name, "%d s_swappc_b64 call site(s)" % k["calls"]))
        if k["flat_atomics"]:
            out.append((name, "%d flat_atomic instruction(s) (generic-pointer atomics)" % k["flat_atomics"]))
        if k["scratch_bytes_per_lane"] > scratch_budget:
            out.append((name, "scratch %d B/lane > budget %d" % (k["scratch_bytes_per_lane"], scratch_budget)))
        if k["vgpr_spill_stores"] > spill_budget:
            out.append((name, "%d VGPR spill stores (%d reloads) > budget %d"
                        % (k["vgpr_spill_stores"], k["vgpr_spill_loads"], spill_budget)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--scratch-budget", type=int, default=128)
    ap.add_argument("--vgpr-spill-budget", type=int, default=0)
    ap.add_argument("--expect-fail", action="store_true", help="succeed only if the gate REJECTS the file")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    kernels, outlined = parse(args.asm)
    if not kernels:
        print("check_kernels: no kernel descriptor in %s" % args.asm)
        return 2
    bad = violations(kernels, outlined, args.scratch_budget, args.vgpr_spill_budget)
    if args.json:
        print(json.dumps({"kernels": kernels, "outlined": outlined, "violations": bad}, indent=1))
    else:
        worst = max(kernels.values(), key=lambda k: k["scratch_bytes_per_lane"])
        print("check_kernels: %d kernels, %d out-of-line functions, max scratch %d B/lane, %d violation(s)"
              % (len(kernels), len(outlined), worst["scratch_bytes_per_lane"], len(bad)))
        for name, what in bad:
            print("  %s: %s" % (name[:100], what))
    if args.expect_fail:
        return 0 if bad else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
