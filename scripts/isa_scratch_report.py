"""Where a kernel's register spills sit: per basic block of the gfx950 ISA, the scratch loads / stores and MFMAs, and whether the
block is a loop body (ends in a backward branch).  Compile-only (no GPU):
    python scripts/isa_scratch_report.py probaforms_amd/csrc/rnvp_mfma_train_nf2.hip 'k_mfma_train<2, 1, 4, 1, 0, true>' ...
Prints, per kernel: scratch instructions inside MFMA-carrying loop bodies (the hot tile loops) against those outside."""
import re, subprocess, sys, tempfile, os

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
                cur = dict(label=m.group(1), start=i, st=0, ld=0, mfma=0, n=0, back=False); blocks.append(cur); continue
            t = l.strip()
            if cur is None or not t or t[0] in ";.":
                continue
            cur["n"] += 1
            cur["st"] += t.startswith("scratch_store"); cur["ld"] += t.startswith("scratch_load"); cur["mfma"] += "v_mfma" in t
            m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m and labels.get(m.group(1), 1 << 30) <= cur["start"]:
                cur["back"] = True
        hot = [b for b in blocks if b["back"] and b["mfma"] >= 8]
        hs = sum(b["st"] + b["ld"] for b in hot)
        allscr = sum(b["st"] + b["ld"] for b in blocks)
        print("%s\n    %d basic blocks, %d instructions, %d MFMAs; scratch instructions: %d in all, %d inside the %d MFMA loop bodies "
              "(%s instructions, %s MFMAs each)" % (name, len(blocks), sum(b["n"] for b in blocks), sum(b["mfma"] for b in blocks), allscr, hs, len(hot),
                                                   "/".join(str(b["n"]) for b in hot), "/".join(str(b["mfma"]) for b in hot)))
        for b in blocks:
            if (b["st"] or b["ld"]) and not (b["back"] and b["mfma"] >= 8):
                print("      outside: %-10s %4d instructions, %2d MFMA, scratch store %2d load %2d" % (b["label"], b["n"], b["mfma"], b["st"], b["ld"]))
        for b in hot:
            if b["st"] or b["ld"]:
                print("      IN LOOP: %-10s %4d instructions, %2d MFMA, scratch store %2d load %2d" % (b["label"], b["n"], b["mfma"], b["st"], b["ld"]))

main()
