"""Cut one kernel's ISA out of a hipcc --save-temps .s file: python scripts/isa_extract.py <file.s> <mangled-name substring> > out.s"""
import sys
src, key = sys.argv[1], sys.argv[2]
on = False
for line in open(src):
    if not on and line.startswith("_ZN") and key in line and line.rstrip().endswith(":") or (not on and line.startswith("_ZN") and key in line and ": ;" in line):
        on = True
    if on:
        sys.stdout.write(line)
        if line.strip().startswith(".end_amdhsa_kernel") or line.strip().startswith("s_endpgm") and False:
            break
        if line.startswith(".Lfunc_end"):
            break
