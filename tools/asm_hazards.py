#!/usr/bin/env python3
"""Static check of the gfx940-family data hazards that hardware does NOT interlock, over the device assembly of the render kernels
(hipcc -S): the compiler pads its own code with s_nop; the hand-written traversal loop (trace_loop_asm in rt_kernels.hip) is
padded by hand, and this script checks both with the same rules -- so a rule that is too strict shows up as a finding in the
compiler's code, and a missing pad in the hand-written loop as a finding between its .Lrt_ labels.

Rules (wait states = instructions issued in between, an `s_nop N` counting N + 1):
  1. a VALU instruction writes an SGPR or VCC (v_cmp*, v_readfirstlane / v_readlane, v_div_scale, carry-outs)  ->  2 wait states before a
     VALU instruction reads that register (as an operand or as a mask)
  2. a VALU instruction writes VCC  ->  4 wait states before v_div_fmas
  3. a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) writes a VGPR  ->  1 wait state before a
     non-transcendental VALU instruction reads it
The scan is linear in program order (labels do not reset it: a fall-through path is a path).
   python tools/asm_hazards.py [file.s]          (without a file: compiles rt_kernels.hip for gfx950 first)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def device_asm(path=None):
    if path:
        return open(path).read()
    out = "/tmp/rt_kernels_hazards.s"
    src = os.path.join(ROOT, "cuda-raytracing_amd", "csrc", "rt_kernels.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                    "--cuda-device-only", "-S", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def regs(tok):
    """register names a token like v3, v[4:7], s[38:39], vcc, |v16|, -v18 stands for"""
    tok = tok.strip().strip("|").lstrip("-")
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return {m.group(1) + m.group(2)}
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return {m.group(1) + str(k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def sgpr_written_by_valu(op, args):
    """SGPRs / VCC a VALU instruction writes"""
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return regs(args[0])
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        return regs(args[0])
    if op.startswith("v_div_scale") or re.match(r"v_(add|sub|subrev)_co_", op) or re.match(r"v_(addc|subb|subbrev)_co_", op) or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
        return regs(args[1]) if len(args) > 1 else set()
    return set()


def check(text):
    findings, kernel, hand = [], None, False
    recent = []                                   # (wait states ago is implicit by position) entries: dict(op, sgpr_w, vgpr_w_trans, line)
    for ln, line in enumerate(text.split("\n"), 1):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kernel, recent, hand = m.group(1), [], False
        if ".Lrt_top" in line and line.strip().endswith(":"):
            hand = True
        if ".Lrt_exit" in line and line.strip().endswith(":"):
            hand = False
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith(".") or kernel is None:
            continue
        parts = code.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if op == "s_endpgm":
            kernel = None
            continue
        if op == "s_nop":
            for e in recent:
                e["age"] += int(args[0]) + 1
            continue
        is_valu = op.startswith("v_")
        if is_valu:
            reads = set()
            dst_n = 2 if (op.startswith("v_div_scale") or re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", op) or re.match(r"v_mad_[ui]64_", op)) else 1
            if op.startswith("v_cmp"):
                dst_n = 1
            for a in args[dst_n:]:
                reads |= regs(a.split(" ")[0])
            # implicit VCC reads of the e32 forms
            if op.startswith("v_cndmask_b32_e32") or op.startswith("v_div_fmas") or re.match(r"v_(addc|subb|subbrev)_co_u32_e32", op):
                reads.add("vcc")
            for e in recent:
                need = 0
                hit = e["sgpr_w"] & reads
                if hit:
                    need = 4 if (op.startswith("v_div_fmas") and "vcc" in hit) else 2
                if e["trans_w"] & reads and not op.startswith(TRANS):
                    need = max(need, 1)
                if need and e["age"] < need:
                    findings.append((kernel, ln, "hand-written loop" if hand else "compiler", "%s (line %d) -> %s: %d wait state(s), %d needed" % (e["op"], e["line"], code, e["age"], need)))
        for e in recent:
            e["age"] += 1
        recent = [e for e in recent if e["age"] < 6]
        if is_valu:
            recent.append({"op": code, "line": ln, "age": 0, "sgpr_w": sgpr_written_by_valu(op, args),
                           "trans_w": regs(args[0]) if op.startswith(TRANS) and args else set()})
    return findings


if __name__ == "__main__":
    f = check(device_asm(sys.argv[1] if len(sys.argv) > 1 else None))
    for k, ln, who, msg in f:
        print("%s:%d [%s] %s" % (k[:60], ln, who, msg))
    print("%d finding(s)" % len(f))
    sys.exit(1 if f else 0)
