#!/usr/bin/env python3
"""Register / scratch report of every kernel in the library (`make -C uvhand_amd/csrc asm` builds the per-file ISA under
build/asm).  A `float4`-valued `cond ? *ptr : zero` once cost the weight-gradient kernel half its speed through a silent spill
(profiles/r01_notes.md), so scratch is pinned PER KERNEL: tools/scratch_allow.json lists every kernel that may use scratch, by
its exact demangled name, with the bytes and the VGPR spills it had when the entry was last reviewed.  Exit code 1 if any kernel
uses scratch and is not listed, or uses MORE than its entry (a regex allow-list with a generous ceiling would let a
new spill inside a loop through).  `--update` rewrites the list from the current build — review the diff before committing it.
What is on it and why: the LDS-stage fused backward calls the dense coarse-level body as a real
function whose prologue saves its callee-saved registers once per workgroup (no spill inside a loop); the deterministic
kept-taps instantiations spill a few registers at the 128-VGPR bound."""
import glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOW = os.path.join(ROOT, "tools", "scratch_allow.json")
shutil.rmtree(os.path.join(ROOT, "build", "asm"), ignore_errors=True)            # no stale listings of files that no longer exist
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "uvhand_amd", "csrc"), "asm"], stderr=subprocess.DEVNULL)
allow = json.load(open(ALLOW)) if os.path.exists(ALLOW) else {}
seen, bad = {}, 0
for f in sorted(glob.glob(os.path.join(ROOT, "build", "asm", "*.s"))):
    s = open(f).read()
    if "amdhsa.kernels" not in s:
        continue
    for blk in s[s.index("amdhsa.kernels"):].split("- .agpr_count:")[1:]:
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        scratch, spill = get("private_segment_fixed_size"), get("vgpr_spill_count")
        flag = ""
        if scratch or spill:
            seen[name] = {"scratch": scratch, "vgpr_spills": spill}
            a = allow.get(name)
            ok = a is not None and scratch <= a["scratch"] and spill <= a["vgpr_spills"]
            flag = "  (scratch: as listed)" if ok else "  <-- SCRATCH %s" % ("not listed" if a is None else "above the listed %d B / %d spills" % (a["scratch"], a["vgpr_spills"]))
            bad += not ok
        print("%-84s vgpr %3d agpr %3s lds %6d scratch %d spills %d%s" % (name[:84], get("vgpr_count"), blk.split("\n")[0].strip(),
                                                                        get("group_segment_fixed_size"), scratch, spill, flag))
if "--update" in sys.argv:
    json.dump(dict(sorted(seen.items())), open(ALLOW, "w"), indent=1)
    print("wrote %s (%d kernels with scratch)" % (ALLOW, len(seen)))
    sys.exit(0)
stale = sorted(set(allow) - set(seen))
if stale:
    print("listed but scratch-free now (drop them with --update):", *stale, sep="\n  ")
sys.exit(1 if bad else 0)
