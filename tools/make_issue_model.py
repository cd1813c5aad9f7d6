"""Builds profiles/r02_issue_model.json — what bench.py's roofline block reads — from
  * rocprofv3 --pmc result files (rocpd SQLite) of ONE workload (instruction-class counters, FETCH_SIZE, WRITE_SIZE),
  * the ISA of the library that was profiled (hipcc -S of the march kernels' translation units, csrc/tu_*.hip, generated here),
  * the measured issue costs (profiles/r02_valu_issue_cycles.json, tools/valu_issue_bench).

usage: python tools/make_issue_model.py --workload "C1 1920x1080x64 guide 16:64 n_gpus 1" --collected "<command>" db1 db2 ...
"""
import argparse
import hashlib
import json
import os
import re
import sqlite3
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import issue_model as im

CSRC = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd", "csrc")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("gpis::", "")


def bench_key(kernel):
    """kernel name as bench.py forms it"""
    k = short(kernel)
    if k.startswith("k_guided_sample_distance") or k.startswith("k_guided_range_sd"):
        return "k_guided_sample_distance"
    if k.startswith("k_guided_transmittance") or k.startswith("k_guided_range_tr"):
        return "k_guided_transmittance"
    if k.startswith("k_fast_sample_distance"):
        return "k_fast_sample_distance"
    if k.startswith("k_fast_transmittance"):
        return "k_fast_transmittance"
    m = re.match(r"k_persist_march<.*, (true|false)>", k)
    if m:
        return "k_persist_march_" + ("sample_distance" if m.group(1) == "true" else "transmittance")
    return None


def isa_symbol(kernel):
    """a substring of the mangled name that identifies the kernel in the ISA text"""
    k = short(kernel)
    m = re.match(r"(k_\w+)<(true|false)>", k)
    if m:
        return "%d%sILb%dE" % (len(m.group(1)), m.group(1), 1 if m.group(2) == "true" else 0)
    m = re.match(r"k_persist_march<(\w+)::Persist, (true|false)>", k)
    if m:
        return "%d%s7PersistELb%dE" % (len(m.group(1)), m.group(1), 1 if m.group(2) == "true" else 0)
    return re.sub(r"<.*", "", k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--collected", default="")
    ap.add_argument("--library", default=os.path.join(CSRC, "libgpis_hip.so"))
    ap.add_argument("--micro", default=os.path.join(ROOT, "profiles", "r02_valu_issue_cycles.json"))
    ap.add_argument("--isa", default=None, help="ISA text of the library (generated with hipcc -S when absent)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_issue_model.json"))
    ap.add_argument("--frames-per-run", type=int, default=0, help="the profiled command rendered this many whole frames per run, some of them "
                    "in several launches (a handle's first frame runs in chunks): counters are then summed per run and divided by this, "
                    "i.e. 'per launch' means per whole-frame launch")
    ap.add_argument("--compulsory", default=None, help="JSON {bench kernel key: compulsory bytes per launch}")
    args = ap.parse_args()

    counters = defaultdict(lambda: defaultdict(list))     # kernel → counter → [per-launch values]
    durations = defaultdict(list)
    for db in args.dbs:
        c = sqlite3.connect(db)
        per_run = defaultdict(lambda: defaultdict(float))
        for name, cn, disp, val in c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1, 2, 3"):
            if args.frames_per_run:
                per_run[name][cn] += val
            else:
                counters[name][cn].append(val)
        for name, cs in per_run.items():
            for cn, tot in cs.items():
                counters[name][cn].append(tot / args.frames_per_run)
        try:
            dur_run = defaultdict(float)
            for name, dur in c.execute("select name, duration from kernels"):
                if args.frames_per_run:
                    dur_run[name] += dur
                else:
                    durations[name].append(dur)
            for name, tot in dur_run.items():
                durations[name].append(tot / args.frames_per_run)
        except sqlite3.Error:
            pass
    isa = args.isa
    if isa is None:
        # the march kernels live in their own translation units (csrc/gpis_launch.hpp): one listing of all of them
        isa = os.path.join(tempfile.gettempdir(), "gpis_hip_model.s")
        with open(isa, "w") as f_out:
            for unit in ("tu_guided_sd.hip", "tu_guided_tr.hip", "tu_fast.hip", "tu_persist_a.hip", "tu_persist_b.hip", "tu_generic_nee.hip"):
                part = isa + "." + unit
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                                       "-I", CSRC, "-S", "--cuda-device-only", "-o", part, os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
                f_out.write(open(part).read())
                os.remove(part)
    costs, _ = im.cost_table(args.micro)
    compulsory = json.load(open(args.compulsory)) if args.compulsory else {}
    sys.path.insert(0, ROOT)
    import bench
    out = {"workload": args.workload, "collected": args.collected, "source_sha256": bench.sources_sha256(),
           "so_sha256_of_the_profiled_build": hashlib.sha256(open(args.library, "rb").read()).hexdigest(),
           "n_simd": im.N_SIMD, "clock_hz_peak": im.PEAK_CLOCK_HZ, "issue_costs": os.path.relpath(args.micro, ROOT), "kernels": {}}
    for kname, cs in counters.items():
        key = bench_key(kname)
        if key is None or "SQ_INSTS_VALU" not in cs:
            continue
        per_launch = {cn: sum(v) / len(v) for cn, v in cs.items()}
        ms = (sum(durations[kname]) / len(durations[kname]) * 1e-6) if durations.get(kname) else None
        blocks = im.parse_kernel_isa(isa, isa_symbol(kname))
        if not blocks:
            print("warning: no ISA for", kname, isa_symbol(kname), file=sys.stderr)
            continue
        mdl = im.model(blocks, costs, per_launch, ms or 1.0)
        entry = {"kernel": short(kname), "launches_profiled": len(cs["SQ_INSTS_VALU"]), "launch_ms_profiled": ms,
                 "valu_insts": per_launch["SQ_INSTS_VALU"], "issue_cycles": mdl["issue_cycles"],
                 "mean_cycles_per_valu_instruction": mdl["mean_cycles_per_valu_instruction"], "fit": mdl["fit"],
                 "unmeasured_mnemonics": mdl["unmeasured_mnemonics"],
                 "frac_at_profiled_duration": mdl["frac"] if ms else None,
                 "counters_per_launch": {k: per_launch[k] for k in sorted(per_launch)}}
        if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
            # rocprofv3 reports KB; on this access pattern (128-B / 96-B records, 4-8-byte guide gathers, scratch) the counters were
            # calibrated against a known byte count in round 1 (ratio 1.007, profiles/README.md): no x2 correction
            entry["traffic_bytes"] = {"fetch": per_launch["FETCH_SIZE"] * 1024.0, "write": per_launch["WRITE_SIZE"] * 1024.0}
        if key in compulsory:
            entry["compulsory_bytes"] = compulsory[key]
        if key in out["kernels"] and out["kernels"][key]["valu_insts"] > entry["valu_insts"]:
            continue            # keep the heavier variant of a kernel pair
        out["kernels"][key] = entry
    json.dump(out, open(args.out, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k, "valu %.3e" % e["valu_insts"], "issue cycles", {a: "%.3e" % b for a, b in e["issue_cycles"].items()}, "ms", e["launch_ms_profiled"], "frac", e["frac_at_profiled_duration"])


if __name__ == "__main__":
    main()
