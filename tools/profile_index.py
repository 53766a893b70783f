"""tools/profile_index.py -- regenerate profiles/INDEX.md: ONE table, kernel -> latest evidence file -> duration -> bytes -> fraction
of the 8 TB/s HBM roof, from (a) the JSON line of a default `python bench.py` run (per-kernel hipEvent durations, algorithmic bytes,
live PMC traffic) and (b) the rocprofv3 summary of the same command (tools/profile_default.sh: --kernel-trace --stats, then
FETCH_SIZE and WRITE_SIZE in their own passes).  Usage: python tools/profile_index.py [tag]     (tag defaults to the newest
profiles/r*_bench_default.json that has a rocprofv3 summary next to it)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def newest_tag():
    tags = []
    for p in glob.glob(os.path.join(PROF, "r*_bench_default.json")):
        t = os.path.basename(p).split("_")[0]
        if os.path.exists(os.path.join(PROF, f"{t}_default_rocprofv3.txt")):
            tags.append(t)
    return sorted(tags)[-1]


def parse_summary(path):
    """kernel-trace rows {short name: (calls, avg_us, min_us, max_us)} and PMC rows {(short name, counter): mean per dispatch}."""
    trace, pmc, mode = {}, {}, None
    for line in open(path):
        if line.startswith("kernel ") and "calls" in line:
            mode = "trace" if not trace else "skip"; continue
        if line.startswith("kernel ") and "counter" in line:
            mode = "pmc"; continue
        if line.startswith("===") or line.startswith("#") or line.startswith("{"):
            continue
        m = re.search(r"(\w+_kernel)(<[^(]*)?", line)
        if not m:
            continue
        name = m.group(1) + (m.group(2) or "")
        name = re.sub(r">\(.*", ">", name).strip()
        if mode == "trace":
            f = line[72:].split()
            if len(f) >= 5:
                trace.setdefault(name, (int(f[0]), float(f[2]), float(f[3]), float(f[4])))
        elif mode == "pmc":
            f = line[60:].split()
            if len(f) >= 3 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                pmc[(name, f[0])] = (int(f[1]), float(f[2]), float(f[4]) if len(f) > 4 else float(f[2]))
    return trace, pmc


def find(table, stem, instantiation=None):
    for k, v in table.items():
        kk = k if isinstance(k, str) else k[0]
        if kk.startswith(stem) and (instantiation is None or instantiation in kk):
            return kk, v
    return None, None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else newest_tag()
    line = json.loads(open(os.path.join(PROF, f"{tag}_bench_default.json")).read().strip().split("\n")[-1])
    summ = f"{tag}_default_rocprofv3.txt"
    trace, pmc = parse_summary(os.path.join(PROF, summ))
    rows = []

    def add(cfg, rl, inst=None):
        if not rl:
            return
        stem = rl["kernel"]
        dur = rl["duration_ms"]
        alg = rl["achieved"] * 1e9 * dur * 1e-3
        name, tr = find(trace, stem, inst)
        # PMC bytes of the SAME launches are only meaningful where every launch of the run is a full sweep: take the per-dispatch
        # maximum (masked launches inside solves read nothing) -- and only where the JSON's own live figure is absent
        fetch = find({k: v for k, v in pmc.items() if k[1] == "FETCH_SIZE"}, stem, inst)[1]
        write = find({k: v for k, v in pmc.items() if k[1] == "WRITE_SIZE"}, stem, inst)[1]
        pmc_bytes = (2 * 1024 * fetch[2] + 1024 * write[2]) if fetch and write else None
        rows.append(dict(cfg=cfg, kernel=name or stem, ev_us=1e3 * dur, tr=tr, alg=alg, frac=rl["frac"], traffic=rl.get("traffic"),
                         frac_traffic=rl.get("frac_traffic"), pmc_bytes=pmc_bytes,
                         bound=rl.get("bound_in_practice", "HBM record stream")[:60]))

    add("C1 (12,4) N=256 x4096 fp64", line["roofline"])
    add("C1", line.get("roofline_forward"))
    def lane_inst(oc, rl):   # the template arguments that tell a shape's instantiation from its neighbours' in the summary
        if not rl:
            return None
        if rl["kernel"].startswith("lane_") or rl["kernel"].startswith("tile32_backward"):
            return f"<{oc['n']}, {oc['m']}"
        if rl["kernel"].startswith("tile32_forward"):
            return f"<{(oc['n'] + 3) // 4}, {(oc['m'] + 3) // 4}"
        return None
    for key, oc in line["config"].get("other_configs", {}).items():
        label = {"c2": "C2 (2,1) N=100 x8192", "c3_8192": "C3 (4,2) N=50 x8192", "c3_65536": "C3 (4,2) N=50 x65536",
                 "c4": "C4 (12,4) N=512 x16384 fp32", "mfma32_13x4": "plan MFMA32 (13,4) N=128 x4096", "mfma32_28x4": "plan MFMA32 (28,4) N=128 x4096"}.get(key, key)
        add(label, oc.get("roofline"), lane_inst(oc, oc.get("roofline")))
        add(label, oc.get("roofline_forward"), lane_inst(oc, oc.get("roofline_forward")))

    out = [f"# profiles/INDEX.md — the current number for every kernel of the hot path, and the file it comes from",
           "",
           f"Regenerated by `python tools/profile_index.py {tag}` from `profiles/{tag}_bench_default.json` (the JSON line of a default",
           f"`python bench.py` run: hipEvent durations inside the timed region, SURVEY §8(d) algorithmic bytes, live PMC traffic) and",
           f"`profiles/{summ}` (rocprofv3 `--kernel-trace --stats` + FETCH_SIZE / WRITE_SIZE passes of the same command, one box).",
           "`frac` = algorithmic bytes ÷ duration ÷ 8 TB/s; `frac_traffic` = PMC bytes (2·1024·FETCH_SIZE + 1024·WRITE_SIZE, gfx950",
           "correction of MI355X_MICROARCH.md) ÷ duration ÷ 8 TB/s. rocprofv3's average mixes the bench's full sweeps with the masked",
           "launches inside the solves the same command runs, so its **max** column is the one to compare with the event duration.",
           "",
           "| config | kernel | events µs (timed region) | rocprofv3 calls / avg / max µs | algorithmic MB per launch | frac | PMC MB per launch (bench live / summary max) | frac_traffic | bound in practice |",
           "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        tr = f"{r['tr'][0]} / {r['tr'][1]:.1f} / {r['tr'][3]:.1f}" if r["tr"] else "–"
        live = f"{r['traffic'] / 1e6:.1f}" if r["traffic"] else "–"
        pm = f"{r['pmc_bytes'] / 1e6:.1f}" if r["pmc_bytes"] else "–"
        ft = f"{r['frac_traffic']:.3f}" if r["frac_traffic"] else "–"
        out.append(f"| {r['cfg']} | `{r['kernel']}` | {r['ev_us']:.1f} | {tr} | {r['alg'] / 1e6:.1f} | {r['frac']:.3f} | {live} / {pm} | {ft} | {r['bound']} |")
    cfg = line["config"]
    out += ["",
            f"Whole-job line of that run: **{line['value'] / 1e6:.2f} M problem-sweeps/s** ({line['ms_per_step']:.3f} ms per sweep of 4096 problems), "
            f"CPU port {line['cpu_baseline']['value']:.0f}/s on one thread" + (f", {line['cpu_baseline']['all_cores']['value'] / 1e3:.1f} k/s on {line['cpu_baseline']['all_cores']['cores']}" if line['cpu_baseline'].get('all_cores') else "") + ".",
            f"`config.ilqr_sweep` {cfg['ilqr_sweep']['ms']:.3f} ms; `config.ilqr_full_solve` {1e3 * cfg['ilqr_full_solve']['seconds']:.2f} ms; "
            f"`config.ilqr_constrained_solve` {1e3 * cfg['ilqr_constrained_solve']['seconds']:.1f} ms ({cfg['ilqr_constrained_solve']['sweeps']} sweeps, "
            f"{cfg['ilqr_constrained_solve']['merit_launches']} merit launches).",
            "",
            "## Every kernel of that command (rocprofv3 --kernel-trace --stats)",
            "",
            "| kernel | calls | avg µs | min µs | max µs |",
            "|---|---|---|---|---|"]
    for k, (calls, avg, mn, mx) in sorted(trace.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:40]:
        out.append(f"| `{k}` | {calls} | {avg:.2f} | {mn:.2f} | {mx:.2f} |")
    out += ["",
            "## Where the other evidence is (latest file per question)",
            "",
            "| question | file | made by |",
            "|---|---|---|"]
    for q, f, t in OTHER:
        mark = "" if os.path.exists(os.path.join(PROF, f)) else " (missing)"
        out.append(f"| {q} | `profiles/{f}`{mark} | `{t}` |")
    open(os.path.join(PROF, "INDEX.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[:24]))


OTHER = [
    ("GPU test suite, last full run", "r04z_gputests.log", "pytest tests -m gpu"),
    ("stream timeline of whole solves on the final kernels: fills, copies, idle", "r04ze_solve_timeline_final.txt", "tools/r04_timeline.sh"),
    ("the same before / after the counter slots (DESIGN 4.17)", "r04g_solve_timeline_after.txt", "tools/r04_timeline.sh"),
    ("a line-search round over the number of problems that take part, both forms (DESIGN 4.19)", "r04p_merit_sparse.txt", "tools/merit_sparse.py"),
    ("instruction / cycle counters of ONE wave in such a round", "r04q_sparse_pmc.txt", "tools/r04_sparse_pmc.sh"),
    ("the constrained merit kernels' counters over whole solves", "r04m_al_merit_pmc.txt", "tools/r04_al_pmc.sh"),
    ("backward / forward / rollout / whole LQ solve for ONE problem", "r04x_backward_sparse.txt", "tools/backward_sparse.py"),
    ("constrained C1 solve and (12,4) MPC example, per kernel, final", "r04s_c1_al_solve.txt", "rocprofv3 --kernel-trace --stats"),
    ("tile-plan instantiations of round 4: data / quadrotor x diagonal / dense x free / bounded", "r04k_tile_variants.txt", "tools/tile_variants.py"),
    ("closed-loop nonlinear MPC of 4096 quadrotors on the tile plan", "r04w_quadrotor_nmpc.txt", "examples/batched_quadrotor_nmpc.py"),
    ("fuzzers at the end of the round (DPP vs LDS forms, fused vs sequenced, GENERIC constraints vs oracle, tile sweeps vs oracle)", "r04zd_fuzz_seeds.txt", "tools/fuzz_*.py, tests/soak/fuzz_*.py"),
    ("HBM bytes per launch, tracked (C1, C4 calibrated; C2, C3 raw)", "pmc_traffic.json", "tools/profile_configs.sh"),
    ("merit pass in DPP form, unconstrained: VALU busy, LDS counters", "r03zb_merit2_pmc.txt", "tools/r03_merit2_pmc.sh"),
    ("C4 kernel variants: ring depth x waves", "r02g_c4_quad_variants.txt", "tools/c4_ab.sh"),
    ("LANE kernels at one wave: instruction and cycle counters", "r01l_lane_pmc_one_wave.txt", "tools/lane_pmc.sh"),
    ("quad vs lane sweeps over the batch size (the policy of DESIGN 4.8)", "r03i_quad_ab.txt", "tools/c3_quad_ab.sh"),
    ("fused vs sequenced solves over the batch size", "r02p_solve_batches.txt", "tools/solve_batches.py"),
    ("both sweeps in one launch (measured, dropped)", "r03y_fused_sweep_experiment.txt", "(experiment, removed)"),
    ("cost of the per-launch events", "r03zd_event_overhead_ext.txt", "tools/event_overhead.py"),
    ("HBM streaming roof of the backward sweep's traffic, no arithmetic", "r01_membench.txt", "tools/membench.hip"),
    ("run-to-run spread on one box", "r01e_bench_repeat.txt", "bench.py x 8"),
]

if __name__ == "__main__":
    main()
