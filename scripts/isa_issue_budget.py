"""Issue-slot budget of a kernel's basic blocks, from the gfx950 ISA (compile-only, no GPU).

    python scripts/isa_issue_budget.py probaforms_amd/csrc/rnvp_mfma_train_nf2.hip 'k_mfma_train<2, 1, 4, 1, 0, true>'

For every basic block that carries MFMAs it counts the instructions by kind and prices them with the measured issue costs of
/opt/skills/guides/MI355X_MICROARCH.md ("Per-instruction cycle constants"): f32-input MFMA 16x16x4 32 cycles (it occupies the SIMD's
f32 data path: VALU work of either wave does not overlap it -- scripts/micro/mfma_valu_overlap.hip), 4x4x1 8 nominal / 10.8 measured
(scripts/micro/mfma4x4.hip), bf16 16x16x32 16 of which 8 block the vector issue, transcendentals 8, other VALU 4, LDS / VMEM / SALU
instructions 4 of the WAVE's issue (they do not hold the SIMD's ALU), s_nop N+1.  The sum over a loop body is the floor for ONE wave
running alone with no stall at all; two waves per SIMD share the ALU (MFMA + VALU columns add) and overlap the rest.
Used for DESIGN.md section 5's slot-by-slot account of k_mfma_train."""
import os, re, subprocess, sys, tempfile

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma"):
        if "4x4x1" in op: return "mfma4"
        if "bf16" in op or "f16" in op: return "mfma_bf16"
        return "mfma16"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op == "s_nop": return "nop"
    if op == "s_waitcnt": return "wait"
    if op == "s_barrier": return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    src, wanted = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                        "--cuda-device-only", "-o", out, src] + os.environ.get("EXTRA", "").split(), check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    for n, (i0, mangled) in enumerate(starts):
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("rnvp::", "")
        if wanted and not any(w in name for w in wanted):
            continue
        i1 = starts[n + 1][0] if n + 1 < len(starts) else len(text)
        lines = text[i0:i1]
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        blocks, cur = [], None
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = dict(label=m.group(1), start=i, back=False, k={}); blocks.append(cur); continue
            t = l.strip()
            if cur is None or not t or t[0] in ";.":
                continue
            c = classify(t)
            cur["k"][c] = cur["k"].get(c, 0) + 1
            if c == "nop":
                cur["k"]["nop_cycles"] = cur["k"].get("nop_cycles", 0) + int(t.split()[1], 0) + 1
            m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m and labels.get(m.group(1), 1 << 30) <= cur["start"]:
                cur["back"] = True
        print(name)
        print("    %-10s %5s | %6s %6s %6s | %5s %5s | %4s %4s %4s %4s %4s | %8s %8s %8s" %
              ("block", "loop", "mfma16", "mfma4", "bf16", "trans", "valu", "lds", "vmem", "salu", "nop", "wait", "ALU cyc", "ALU@10.8", "wave cyc"))
        tot = {}
        for b in blocks:
            k = b["k"]
            nm = k.get("mfma16", 0) + k.get("mfma4", 0) + k.get("mfma_bf16", 0)
            for key, v in k.items():
                tot[key] = tot.get(key, 0) + v
            if nm < 4:
                continue
            alu = 32 * k.get("mfma16", 0) + 8 * k.get("mfma4", 0) + 8 * k.get("mfma_bf16", 0) + 8 * k.get("trans", 0) + 4 * k.get("valu", 0)
            alu2 = alu + 2.8 * k.get("mfma4", 0)
            wave = alu2 + 8 * k.get("mfma_bf16", 0) + 4 * (k.get("lds", 0) + k.get("vmem", 0) + k.get("salu", 0)) + k.get("nop_cycles", 0)
            print("    %-10s %5s | %6d %6d %6d | %5d %5d | %4d %4d %4d %4d %4d | %8d %8d %8d" %
                  (b["label"], "yes" if b["back"] else "", k.get("mfma16", 0), k.get("mfma4", 0), k.get("mfma_bf16", 0), k.get("trans", 0),
                   k.get("valu", 0), k.get("lds", 0), k.get("vmem", 0), k.get("salu", 0), k.get("nop_cycles", 0), k.get("wait", 0), alu, alu2, wave))
        print("    whole kernel (static):", {k: v for k, v in sorted(tot.items())})


main()
