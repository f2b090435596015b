#!/usr/bin/env python3
"""Static instruction mix of the loops of one gfx950 kernel, from the built object (no GPU needed).

usage: python tools/loop_instructions.py algoplonk_amd/csrc/backend_bn254.o ntt_pass_kernel [--min 40]

Unbundles the gfx950 code object from the .o (objcopy .hip_fatbin + clang-offload-bundler), disassembles it with llvm-objdump,
takes the first kernel whose mangled name contains the pattern, finds its loops (a branch to a lower address closes one) and
prints, per loop body, how many instructions of each class it holds.  Loop bodies here are straight-line apart from skipped
exec-masked regions, so the count of a body is what one trip issues per wave when every region is taken.

This is how DESIGN.md's "374 instructions around a 215-instruction product" is broken down: the classes
are  mad64 = v_mad_u64_u32, valu = every other v_* instruction, lds = ds_*, vmem = global_* / buffer_* / flat_*, salu = s_*
(s_waitcnt / s_nop counted apart as wait).
"""
import argparse
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="loopins_")
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def kernel_lines(text, pattern):
    out, on = [], False
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
        if m:
            if on:
                break
            if pattern in m.group(2):
                on = True
                name = m.group(2)
                base = int(m.group(1), 16)
            continue
        if on and ln.startswith("\t"):
            out.append(ln)
    if not on:
        raise SystemExit("no kernel matching %r" % pattern)
    return name, base, out


def classify(op):
    if op == "v_mad_u64_u32":
        return "mad64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop"):
        return "wait"
    return "salu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("pattern")
    ap.add_argument("--min", type=int, default=40, help="skip loops with fewer VALU instructions than this")
    ap.add_argument("--ops", action="store_true", help="also list the VALU opcodes of each loop")
    a = ap.parse_args()
    name, base, lines = kernel_lines(disassemble(a.obj), a.pattern)
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    ins = []   # (addr, opcode, branch_target or None)
    for ln in lines:
        m = re.match(r"^\t(\S+)(.*?)//\s*([0-9A-Fa-f]+):", ln)
        if not m:
            continue
        op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"\+0x([0-9a-f]+)>", ln)
            if t:
                tgt = base + int(t.group(1), 16)
        ins.append((addr, re.sub(r"_e(32|64)$", "", op), tgt))
    print("kernel: %s" % re.sub(r"\(.*", "", demangled))
    print("instructions: %d (mad64 %d)" % (len(ins), sum(1 for i in ins if i[1] == "v_mad_u64_u32")))
    loops = sorted({(t, ad) for ad, op, t in ins if t is not None and t <= ad})
    # keep innermost-first order by size
    print("%-22s %6s %6s %6s %5s %5s %5s %5s   %s" % ("loop [start, end]", "total", "mad64", "valu", "lds", "vmem", "salu", "wait", "VALU all"))
    for t, ad in sorted(loops, key=lambda x: x[1] - x[0]):
        body = [i for i in ins if t <= i[0] <= ad]
        c = {}
        for _, op, _ in body:
            c[classify(op)] = c.get(classify(op), 0) + 1
        valu_all = c.get("mad64", 0) + c.get("valu", 0)
        if valu_all < a.min:
            continue
        print("[+0x%05x, +0x%05x] %6d %6d %6d %5d %5d %5d %5d   %d" % (t - base, ad - base, len(body), c.get("mad64", 0), c.get("valu", 0),
                                                                  c.get("lds", 0), c.get("vmem", 0), c.get("salu", 0), c.get("wait", 0), valu_all))
        if a.ops:
            ops = {}
            for _, op, _ in body:
                if op.startswith("v_"):
                    ops[op] = ops.get(op, 0) + 1
            print("    " + ", ".join("%s %d" % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
