#!/usr/bin/env python3
"""VALU lane-instructions ONE lane needs for ONE event of the path, counted in the gfx950 ISA (no GPU needed).

  python tools/isa_event_costs.py            # writes profiles/isa_event_costs.json

tools/micro/event_costs.hip holds one kernel per event (ray set-up, box test, quad / sphere test, one shade per material kind, a miss,
one primary ray) built from the SAME device functions as the product (rt_path.h, rt_device.h) and a baseline kernel with the same loads
and stores; this script compiles it with the library's flags, counts the v_* instructions between each kernel's label and its end, and
subtracts the baseline.  The hand-written box-step loops are counted in their source text (rt_path.h box_loop_flat / _lds / _compact:
vector instructions per box step).  bench.py multiplies the table by the event counts of its counting pass: `roofline.useful_frac`
= lane-instructions the path's arithmetic needs per second / the chip's lane-slots per second.
Static counts: where the compiler kept a branch both sides are counted once (range reductions of sin / cos, the NaN guards of the short
division), loops once per iteration (the unit-disk rejection loop: expected 4 / pi iterations, taken into account by bench.py)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tiny-raytracer_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]        # csrc/Makefile CXXFLAGS (code generation part)


def kernel_source_digest():
    """bench.py's digest of the kernel sources (comments stripped): one function, imported, so that the two can never disagree."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_for_digest", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.kernel_source_digest()


def count_kernels(asm_text):
    out = {}
    for m in re.finditer(r"^(ev_\w+):.*?\n(.*?)^\.Lfunc_end\d+:", asm_text, re.S | re.M):
        ins = []
        for line in m.group(2).splitlines():
            t = line.strip()
            if not t or t.startswith((".", ";", "//")) or t.split()[0].endswith(":"):
                continue
            ins.append(t.split()[0])
        out[m.group(1)] = {"valu": sum(i.startswith("v_") for i in ins), "salu": sum(i.startswith("s_") for i in ins),
                           "vmem": sum(i.startswith(("global_", "buffer_", "flat_", "scratch_")) for i in ins),
                           "trans": sum(i.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")) for i in ins)}
    return out


def asm_loop_counts():
    """Vector instructions per box step of the three hand-written loops, from the text of their asm blocks."""
    src = open(os.path.join(CSRC, "rt_path.h")).read()

    def vcount(text):                         # mnemonics inside the string literals only (comments name instructions too)
        return sum(len(re.findall(r"\bv_[a-z0-9_]+", lit)) for lit in re.findall(r'"((?:[^"\\]|\\.)*)"', text))

    flat_box = re.search(r"#define TRT_FLAT_BOX\(.*?\n((?:.*\\\n)*.*\n)", src).group(0)
    per_box = vcount(flat_box)                                              # the slab test proper
    # box_loop_flat: + v_mov (link copy) and v_add (stack top) under the push mask per box
    flat = per_box + 2
    def trip(name):                       # the loop body proper: between the labels 1: and 2: of the asm block
        block = re.search(r"TRT_DEV float2\* " + name + r"\(.*?asm volatile\((.*?)\n\s*:", src, re.S).group(1)
        return block[block.index('"1:'):block.index('"2:')]
    lds, compact = trip("box_loop_lds"), trip("box_loop_compact")
    return {"box_step_flat": flat, "box_step_lds": vcount(lds), "box_step_compact": vcount(compact), "slab_test_alone": per_box}


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, "-c", os.path.join(ROOT, "tools", "micro", "event_costs.hip"),
               "-save-temps=obj", "-o", os.path.join(tmp, "ev.o")]
        subprocess.run(cmd, check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        raw = count_kernels(open(os.path.join(tmp, asm[0])).read())
    base = raw["ev_baseline"]["valu"]
    table = {k[3:]: v["valu"] - base for k, v in raw.items() if k != "ev_baseline"}
    out = {"_about": "VALU instructions one lane needs for one event (gfx950 ISA of tools/micro/event_costs.hip minus its baseline kernel; "
                     "box_step_*: vector instructions per trip of the hand-written loops in rt_path.h). Made by tools/isa_event_costs.py.",
           "kernel_source_digest": kernel_source_digest(), "baseline_valu": base, "events": table, "asm_loops": asm_loop_counts(),
           "raw": raw}
    path = os.path.join(ROOT, "profiles", "isa_event_costs.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("events", "asm_loops", "kernel_source_digest")}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
