#!/usr/bin/env python3
"""Register / scratch report of every kernel in the library (`make -C uvhand_amd/csrc asm` first builds the
per-file ISA under build/asm).  Exit code 1 if any kernel spills or uses scratch memory that is not on the list below: a
`float4`-valued `cond ? *ptr : zero` once cost the weight-gradient kernel half its speed that way (profiles/r01_notes.md).
Expected: the LDS-stage fused backward calls the dense coarse-level body as a real function, whose
prologue saves its callee-saved registers to scratch once per workgroup (no VGPR spill inside any loop); the deterministic
kept-taps instantiations spill a few registers at the 128-VGPR bound."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "uvhand_amd", "csrc"), "asm"])
bad = 0
for f in sorted(glob.glob(os.path.join(ROOT, "build", "asm", "*.s"))):
    s = open(f).read()
    if "amdhsa.kernels" not in s:
        continue
    for blk in s[s.index("amdhsa.kernels"):].split("- .agpr_count:")[1:]:
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        scratch, spill = get("private_segment_fixed_size"), get("vgpr_spill_count")
        expected = ("bwd_fused_lds_d32_kernel" in name and scratch <= 192 and spill <= 16) or \
                   ("bwd_fused_d32_kernel" in name and re.search(r"<\d, 3, .*, true>$", name) and scratch <= 64)      # kAccWide + DET
        flag = ("  (scratch: expected)" if expected else "  <-- SCRATCH") if (scratch or spill) else ""
        bad += bool(flag) and not expected
        print("%-84s vgpr %3d agpr %3s lds %6d scratch %d%s" % (name[:84], get("vgpr_count"), blk.split("\n")[0].strip(),
                                                              get("group_segment_fixed_size"), scratch, flag))
sys.exit(1 if bad else 0)
