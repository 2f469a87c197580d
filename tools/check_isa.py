#!/usr/bin/env python3
"""Static check of the gfx950 ISA of the HIP core (no GPU needed).

Compiles basal_core.hip to assembly and fails if a kernel contains a label whose next instructions
branch straight back to it on a loop-invariant condition: that shape is what hipcc (ROCm 7.2)
emitted when jump threading split lane 0 from lanes 1..63 across the persistent work loop (see
lane0() in basal_core.hip) -- on the GPU it showed up as a hang or as reads aligned by a partial
wave.  Also reports VGPR/SGPR-spill/scratch/LDS use per kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "basal_amd", "csrc", "basal_core.hip")


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "core.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S",
                            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", SRC, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            return 2
        lines = open(out).read().split("\n")
    bad = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        lab = re.escape(m.group(1))
        for j in range(i + 1, min(i + 4, len(lines))):
            if re.search(r"s_c?branch\w*\s+" + lab + r"\s*$", lines[j]):
                print("self-loop at asm line %d: %s" % (i + 1, m.group(1)))
                bad += 1
                break
    name = None
    usage = {}
    for l in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            name = m.group(1)
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", l)
        if m and name:
            print("%-60s %-28s %s" % (name[-60:], m.group(1), m.group(2)))
            usage.setdefault(name, {})[m.group(1)] = int(m.group(2))
    # LDS is handed out in units of 1 280 B on gfx950 (160 KB = 128 units per CU): a block of four waves one unit over costs a whole block per CU.
    # (Measured, round 4: 8 bytes more per wave took align_kernel<4,*,GAP> from 25 to 26 units, from five blocks per CU to four, 101 -> 114 ms.)
    # The compiler's occupancy figure (registers) is what the launch code counts on: blocks per CU = waves per SIMD for 256-thread blocks.
    for name, u in usage.items():
        if "align_kernel" not in name or "LDS Size [bytes/block]" not in u or "Occupancy [waves/SIMD]" not in u:
            continue
        units = -(-u["LDS Size [bytes/block]"] // 1280)
        m = re.search(r"align_kernelILi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
        want = u["Occupancy [waves/SIMD]"]
        if m:  # the HEAVY GAP kernels and the 8-/16-word GAP kernels are launched with fewer waves than their registers allow (waves_per_simd)
            nwt, gap, heavy = int(m.group(1)), m.group(3) == "1", m.group(4) == "1"
            if gap and heavy:
                want = min(want, {4: 4, 8: 3, 16: 2}[nwt])
        if units * want > 128:
            print("LDS: %s takes %d units of 1280 B per block, %d blocks per CU need %d > 128" % (name[-60:], units, want, units * want))
            bad += 1
    # a device function that was NOT inlined into its kernel (it then takes LDS and global pointers as generic ones: flat loads, a stack of
    # several hundred bytes per lane) -- every function of this file is meant to end up inside an align_kernel instantiation or a small kernel
    for l in lines:
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+(process_read|heavy_mode|heavy_flush|bulk_add|bulk_add2|gap_flush|gap_search|dups_among|dups_stored|stream_flush|prep_read|reorder_seed|add_hit|gap_align)\w*):", l)
        if m:
            print("not inlined: %s" % m.group(1)[:100])
            bad += 1
    if re.search(r"\bflat_(load|store)", "\n".join(lines)):
        print("flat_* memory instructions present (LDS/global address space not resolved)")
        bad += 1
    print("ISA check: %s" % ("FAILED" if bad else "ok"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
