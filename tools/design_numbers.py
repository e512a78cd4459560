#!/usr/bin/env python3
"""design_numbers.py <tag> [--write] -- the block of measured numbers DESIGN.md quotes, regenerated from the tracked profile files of one
round (profiles/<tag>_*: kernel traces, bench lines, PMC summaries, stage cycles) so that a share or a time in the design text is the
one in the profile. Prints the block; with --write replaces the text between the markers
    <!-- numbers:begin --> ... <!-- numbers:end -->
of DESIGN.md with it."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def read(name):
    path = os.path.join(P, name)
    return open(path).read() if os.path.exists(path) else ""


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def trace_rows(text):
    rows = []
    for line in text.splitlines()[1:]:
        m = re.match(r"\s*([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if not m:
            continue
        name = m.group(5).replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", name).strip()
        rows.append((float(m.group(1)), int(m.group(2)), float(m.group(4)), name))
    return rows


def short_kernel(name):
    name = re.sub(r"^vio_wk::", "", name)
    return name if len(name) < 60 else name[:57] + "..."


def stage_line(text, which):
    for line in text.splitlines():
        if line.startswith("stage cycles") and which in line:
            return dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)\(", line))
    return {}


def pmc_counters(text, prefix=""):
    out = {}
    for line in text.splitlines():
        if prefix and not line.startswith(prefix):
            continue
        if not prefix and line.startswith("max_iter"):
            continue
        m = re.search(r"(SQ\w+|TCC\w+|SQC\w+)\s+calls=\s*\d+\s+avg=\s*([\d.]+).*full_batch_avg=\s*([\d.]+)", line)
        if m:
            out[m.group(1)] = float(m.group(3))
    return out


GROUPS = [
    ("band + pose factorization (chain wave's clock)", ["d3", "c_ahead", "c_wait", "c_potrf", "c_trsm", "cholesky"]),
    ("projection factors: evaluation, staging, Gram products", ["p_zero", "p_fact", "p_gram", "p_feat", "eval_proj", "e_head"]),
    ("IMU factors (raw evaluation + matrix-core products)", ["imu_raw", "eval_imu"]),
    ("prior (dx, H0 dx, cost, gradient)", ["eval_prior"]),
    ("Schur complement + right-hand side, diagonal scaling", ["rhs", "schur", "scale", "tr_vec"]),
    ("back-substitution (pose tiles, band chains, landmarks)", ["tri_solve", "backsolve", "b_init", "b_pose", "b_asp", "b_band", "b_gn", "x0"]),
    ("trust-region vector phases (Cauchy point, dogleg, Plus, norms)", ["quad_form", "q_w", "dogleg", "v_gd", "v_dot", "v_step", "v_plus", "v_gmax", "v_rest"]),
    ("cost-only evaluations of rejected candidates", ["cost_eval", "d0", "d1", "d2"]),
    ("set-up (cov^-1, H0 = J0^T J0)", ["setup_imu", "setup_prior"]),
    ("new2old + marginalization", ["new2old", "marg_build", "marg_chol", "m_prior", "m_imu", "m_fact", "m_gram"]),
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r06_a"
    out = []
    b = last_json(read(tag + "_bench.json"))
    if b:
        km = b["config"].get("kernel_ms", {})
        r = b["roofline"]
        out.append("**Bench line** (`profiles/%s_bench.json`, `python bench.py`, %s sequences per GPU, %d steps): **%.1f k frames/s**, "
                   "%.3f ms per step = %.3f ms front-end + %.3f ms window kernel; `roofline.frac` **%.4f** (%.2f TFLOP/s of %.1f fp64), "
                   "`roofline.traffic` %.2f GB per launch; `roofline.frontend.frac` %.3f by SURVEY 8(d)'s formula, **%.3f at the measured %.2f LK "
                   "iterations**; `cpu_baseline` %.1f frames/s on %d core (%s)." % (
                       tag, b["config"].get("sequences_per_gpu", "?"), b["steps"], b["value"] / 1e3, b["ms_per_step"], km.get("frontend_step", 0),
                       km.get("window_solve", 0), r["frac"], r["achieved"], r["peak"], (r.get("traffic") or 0) / 1e9,
                       r.get("frontend", {}).get("frac", 0), r.get("frontend", {}).get("frac_at_measured_lk_iterations", 0),
                       r.get("frontend", {}).get("lk_mean_iterations", 0), b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"],
                       "front-end %.1f ms + reference Ceres solve %.1f ms per frame" % (b["cpu_baseline"].get("frontend_ms", 0), b["cpu_baseline"].get("solve_ms", 0))))
        sec = b["config"].get("secondary")
        if sec:
            out.append("")
            out.append("| secondary leg (`config.secondary`) | value |")
            out.append("|---|---|")
            for k in ("configs2", "configs4", "configs2_256", "configs4_256"):
                if k in sec and isinstance(sec[k], dict):
                    s = sec[k]
                    out.append("| `%s` | %.1f k frames/s; window solve %.2f ms, front-end %.2f ms per step; `frac` %.3f, front-end `frac` %.3f |" % (
                        k, s["frames_per_s"] / 1e3, s["window_ms"], s["frontend_ms"], s["frac"], s["frac_frontend"]))
            if "end_to_end_full" in sec:
                e = sec["end_to_end_full"]
                out.append("| `end_to_end_full` (host frames in, host states out, asynchronous submit) | %s camera frames/s at 256 / 512 sequences; app cadence (FREQ 3) %s at 256;" % (
                    " / ".join("%.1f k" % (e[k] / 1e3) for k in ("256", "512") if k in e), "%.1f k" % (e.get("freq3_256", 0) / 1e3))
                    + (" **%.1f k at 512 with the frame buffers registered** (`vio_host_register`) |" % (e["512_registered_host_frames"] / 1e3)
                       if e.get("512_registered_host_frames") else " |"))
            if "end_to_end_solves_per_s" in sec:
                out.append("| `end_to_end` (estimator path, 512 sequences) | %.1f k window solves/s |" % (sec["end_to_end_solves_per_s"] / 1e3))
            if "resident_256" in sec:
                s = sec["resident_256"]
                out.append("| `resident_256` | %.1f k frames/s, window kernel %.3f ms, `frac` %.3f |" % (s["frames_per_s"] / 1e3, s["window_ms"], s["frac"]))
            if "small_batches_ms" in sec:
                s = sec["small_batches_ms"]
                out.append("| `small_batches` | %s ms for 1 / 8 / 64 windows per launch |" % " / ".join("%.2f" % s[k] for k in ("B1", "B8", "B64") if k in s))
            if "frontend_as_tracked" in sec:
                s = sec["frontend_as_tracked"]
                out.append("| `frontend_as_tracked` (no detection for sequences that still track MAX_CNT features) | %.1f k frames/s, front-end step %.2f ms |" % (
                    s["frames_per_s"] / 1e3, s["frontend_ms"]))
    kt = trace_rows(read(tag + "_kernel_trace.txt"))
    if kt:
        out.append("")
        out.append("**Kernel trace of the bench step** (`profiles/%s_kernel_trace.txt`, `rocprofv3 --kernel-trace --stats`):" % tag)
        out.append("")
        out.append("| kernel | share | calls | average µs |")
        out.append("|---|---|---|---|")
        for pct, calls, avg, name in kt:
            if pct >= 0.5:
                out.append("| `%s` | %.1f %% | %d | %.1f |" % (short_kernel(name), pct, calls, avg))
    for leg in ("configs2", "configs4"):
        rows = trace_rows(read("%s_kernel_trace_%s.txt" % (tag, leg)))
        if rows:
            top = [r_ for r_ in rows if r_[0] >= 2.0]
            out.append("")
            out.append("`profiles/%s_kernel_trace_%s.txt`: " % (tag, leg) + "; ".join("`%s` %.0f µs (%.0f %%)" % (short_kernel(n), a, p_) for p_, c, a, n in top))
    st = read(tag + "_solver_stage_cycles.txt")
    for which, label in (("window 0 of 1)", "one window alone"), ("window 0 of 512)", "window 0 of a 512-window launch (two per CU)")):
        cyc = stage_line(st, which)
        if not cyc:
            continue
        tot = max(1, cyc.get("total", 1))
        out.append("")
        out.append("**Stage clock, %s** (`profiles/%s_solver_stage_cycles.txt`, the profiling instantiation; %.2f M cycles):" % (label, tag, tot / 1e6))
        out.append("")
        out.append("| group | share | stages (k cycles) |")
        out.append("|---|---|---|")
        seen = set()
        for name, keys in GROUPS:
            have = [(k, cyc[k]) for k in keys if cyc.get(k, 0) > 0]
            seen.update(keys)
            if have:
                out.append("| %s | %.1f %% | %s |" % (name, 100.0 * sum(c for _, c in have) / tot, ", ".join("`%s` %.0f" % (k, c / 1e3) for k, c in have)))
        rest = [(k, c) for k, c in cyc.items() if k not in seen and k != "total" and c > 0]
        if rest:
            out.append("| other | %.1f %% | %s |" % (100.0 * sum(c for _, c in rest) / tot, ", ".join("`%s` %.0f" % (k, c / 1e3) for k, c in rest)))
    sq = read(tag + "_pmc_sq.txt")
    c = pmc_counters(sq)
    if c:
        wc = c.get("SQ_WAVE_CYCLES", 0)
        out.append("")
        out.append("**Counters of the window kernel, per 512-window launch** (`profiles/%s_pmc_sq.txt`, one `--pmc` pass per group): " % tag +
                   "`SQ_WAVE_CYCLES` %.0f M, `SQ_WAIT_ANY` %.1f %%, `SQ_WAIT_INST_ANY` %.1f %%, `SQ_ACTIVE_INST_ANY` %.1f %% (VALU %.1f %%, scalar %.1f %%, LDS %.1f %%); "
                   "per window %.0f k VALU + %.0f k SALU + %.1f k LDS + %.1f k VMEM + %.1f k MFMA wave-instructions; matrix pipes busy %.0f M SIMD-cycles; "
                   "L2: %.1f M requests, %.0f %% hits." % (
                       wc / 1e6, 100 * c.get("SQ_WAIT_ANY", 0) / max(wc, 1), 100 * c.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1),
                       100 * c.get("SQ_ACTIVE_INST_ANY", 0) / max(wc, 1), 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(wc, 1),
                       100 * c.get("SQ_ACTIVE_INST_SCA", 0) / max(wc, 1), 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / max(wc, 1),
                       c.get("SQ_INSTS_VALU", 0) / 512e3, c.get("SQ_INSTS_SALU", 0) / 512e3, c.get("SQ_INSTS_LDS", 0) / 512e3,
                       c.get("SQ_INSTS_VMEM", 0) / 512e3, c.get("SQ_INSTS_MFMA", 0) / 512e3, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1e6,
                       c.get("TCC_REQ_sum", 0) / 1e6, 100 * c.get("TCC_HIT_sum", 0) / max(c.get("TCC_REQ_sum", 1), 1)))
        c2 = pmc_counters(sq, "max_iter=2")
        if c2 and "SQ_INSTS_VALU" in c2:
            out.append("Per trust-region iteration (the same counters at 2 and at 10 iterations, difference / 8): **%.1f k VALU + %.1f k SALU + %.1f k LDS + "
                       "%.2f k MFMA** wave-instructions per window." % tuple((c[k] - c2[k]) / 8 / 512e3 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA")))
    pj = read(tag.split("_")[0] + "_pmc.json")
    if pj:
        d = json.loads(pj)
        w = d.get("vio_window_kernel", {})
        if w:
            out.append("")
            out.append("**HBM-side traffic** (`profiles/%s_pmc.json`; FETCH_SIZE x %.3f + WRITE_SIZE, factors from `tools/microbench/pmc_calib`): %.2f GB per 512-window launch = "
                       "**%.2f MB per window** (%.2f MB read, %.2f MB written)." % (
                           tag.split("_")[0], d["calibration"]["read8_kernel"]["factor"], w["bytes_per_launch"] / 1e9, w["bytes_per_launch"] / 512e6,
                           w["fetch_reported_bytes"] * d["calibration"]["read8_kernel"]["factor"] / 512e6, w["write_reported_bytes"] / 512e6))
    lw = read(tag + "_large_windows.txt")
    if lw:
        ks = re.findall(r"(configs\[\d\]) B=(\d+) kernel ([\d.]+) ms", lw)
        if ks:
            out.append("")
            out.append("**Large windows, kernel only** (`profiles/%s_large_windows.txt`): " % tag + "; ".join("%s x %s: %s ms" % k for k in ks) + ".")
    block = "\n".join(out)
    print(block)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        a, e = "<!-- numbers:begin -->", "<!-- numbers:end -->"
        if a in s and e in s:
            s = s[:s.index(a) + len(a)] + "\n" + block + "\n" + s[s.index(e):]
            open(path, "w").write(s)
            print("\n[DESIGN.md updated]", file=sys.stderr)
        else:
            print("\n[markers not found in DESIGN.md]", file=sys.stderr)


if __name__ == "__main__":
    main()
