#!/usr/bin/env python3
"""Static VALU issue-cost model of the library's kernels from their gfx950 ISA and the measured per-instruction issue
rates (tools/ubench/valu_rate.hip -> profiles/r03_valu_rate.json).

On MI355X a wave64 VALU instruction occupies its SIMD for ~2.25 clocks if it is one of the "fast" f32 / simple integer
ops (v_add/sub/mul/fma_f32, v_add_u32, v_and/or/xor, v_mov_b32, shifts) and ~4.2-4.4 clocks otherwise (every compare,
v_cndmask, min/max/med3, v_bfi, packed f32, all f64 and 64-bit integer ops), see profiles/r03_valu_rate.md.  The PMC
counters cannot tell the two apart (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU for both), so the busy fraction of a kernel is
estimated as  SQ_INSTS_VALU x (static mix-weighted clocks per instruction) / (SIMDs x clock x launch time).
The mix is STATIC (instruction counts in the kernel body, loops not weighted): an estimate, stated as such.

usage: python tools/valu_model.py [out.json]     (compiles msfl_api.hip to ISA with hipcc -S: ~30 s, no GPU needed)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32|^v_(add|sub|subrev)_(u32|i32|co_u32)|^v_(and|or|xor|not)_b32|^v_mov_b32|"
                  r"^v_(lshlrev|lshrrev|ashrrev)_(b32|i32)|^v_(and_or|or3|lshl_or|lshl_add|add3|add_lshl|xad)_(b32|u32)|^v_cvt_f32_(i32|u32)|^v_accvgpr")
SLOW_TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_|^v_mul_(lo|hi)_|^v_div_|^v_mad_(u64|i64)")
CLK_FAST, CLK_4, CLK_TRANS = 2.25, 4.3, 8.6      # profiles/r03_valu_rate.md; transcendental / 32x32 multiplies taken at half the 4-clock rate


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):                   # not the first s_endpgm: a kernel with an early exit has several
            yield name, body
            name = None
            continue
        m = re.match(r"^([a-z_0-9]+)", t)
        if m and not t.startswith((".", ";")):
            body.append(m.group(1))


def main():
    import argparse
    ap = argparse.ArgumentParser(description="static VALU instruction mix of every msfl kernel (ISA of the gfx950 build)")
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "profiles", "r03_valu_mix.json"), help="JSON file to write")
    out = ap.parse_args().out
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "dev.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w", "-S", "--cuda-device-only",
                               os.path.join(ROOT, "msf_loam_amd", "csrc", "msfl_api.hip"), "-o", s])
        asm = open(s).read()
    res = {}
    for name, body in kernels(asm):
        if "msfl" not in name or "rocprim" in name:
            continue
        valu = [i for i in body if i.startswith("v_")]
        if not valu:
            continue
        fast = sum(1 for i in valu if FAST.match(i))
        trans = sum(1 for i in valu if SLOW_TRANS.match(i))
        four = len(valu) - fast - trans
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "").replace("msfl::", "")
        res[demangled] = {"valu_static": len(valu), "fast_2clk": fast, "four_clk": four, "slow": trans,
                          "clocks_per_inst_static_mix": (fast * CLK_FAST + four * CLK_4 + trans * CLK_TRANS) / len(valu),
                          "salu_static": sum(1 for i in body if i.startswith("s_")), "vmem_static": sum(1 for i in body if i.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                          "lds_static": sum(1 for i in body if i.startswith("ds_"))}
    json.dump({"clocks": {"fast": CLK_FAST, "four": CLK_4, "slow": CLK_TRANS}, "source": "tools/valu_model.py (static ISA mix) + profiles/r03_valu_rate.json",
               "kernels": res}, open(out, "w"), indent=1)
    for k in sorted(res, key=lambda k: -res[k]["valu_static"])[:14]:
        r = res[k]
        print("%-60s VALU %5d  fast %4.0f %%  4-clk %4.0f %%  slow %3.0f %%  -> %.2f clk/inst" % (k[:60], r["valu_static"], 100 * r["fast_2clk"] / r["valu_static"],
              100 * r["four_clk"] / r["valu_static"], 100 * r["slow"] / r["valu_static"], r["clocks_per_inst_static_mix"]))


if __name__ == "__main__":
    main()
