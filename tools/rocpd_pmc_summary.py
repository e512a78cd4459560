#!/usr/bin/env python3
"""Per-kernel averages of PMC counters from rocprofv3's rocpd databases (separate --pmc passes, tools/profile_round.sh).

    tools/rocpd_pmc_summary.py <results.db> [...]                     text: one line per (kernel, counter)
    tools/rocpd_pmc_summary.py --json out.json --workload "<key>" --calib <calib_fetch.db> <calib_write.db> \
                               --fetch <bench_fetch.db> --write <bench_write.db>
        -> the summary bench.py reads for roofline.traffic: bytes per launch of the window kernel and of one front-end step,
           FETCH_SIZE / WRITE_SIZE scaled by the factors measured on tools/microbench/pmc_calib (known 1 GiB streams).
"""
import argparse
import json
import sqlite3
import sys

# kernels of one front-end step (bench.py's `roofline_frontend` label names the same set), matched by base name: the
# demangled names carry template arguments (lk_track_kernel<1>, detect_kernel<false>)
FRONTEND = ["copy_frames_kernel", "pyr_down_kernel", "pyr_down4_kernel", "lk_track_kernel", "track_update_kernel", "detect_kernel", "corner_select_kernel"]
ONCE_PER_STEP = "corner_select_kernel"   # every bench step publishes: its dispatch count is the number of steps


def short(name):
    """'void (anonymous namespace)::lk_track_kernel<1>(unsigned char const*, ...)' -> 'lk_track_kernel<1>'."""
    name = name.replace("(anonymous namespace)::", "")
    depth, end = 0, len(name)
    for i, ch in enumerate(name):  # the argument list opens at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    head = name[:end]
    cut = head.index("<") if "<" in head else len(head)
    sp, ns = head.rfind(" ", 0, cut), head.rfind("::", 0, cut)
    return head[max(sp + 1, ns + 2 if ns >= 0 else 0):]


def base(name):
    return name.split("<")[0]


def load(path):
    """-> {(kernel, counter): [per-dispatch values]}"""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
                      "group by kernel_name, counter_name, dispatch_id").fetchall()
    agg = {}
    for k, c, _, v in rows:
        agg.setdefault((short(k), c), []).append(v)
    return agg


def by_base(agg, kernel_base, counter):
    """All dispatches of every instantiation of a kernel."""
    out = []
    for (k, c), vs in agg.items():
        if c == counter and base(k) == kernel_base:
            out.extend(vs)
    return out


def text(paths):
    for path in paths:
        for (k, c), vs in sorted(load(path).items(), key=lambda t: -sum(t[1])):
            big = [x for x in vs if x >= 0.8 * max(vs)] or vs  # full-batch launches (the run also has 1-window warm-ups)
            print("%-28s %-22s calls=%4d avg=%14.1f min=%14.1f max=%14.1f full_batch_avg=%14.1f" % (k, c, len(vs), sum(vs) / len(vs), min(vs), max(vs), sum(big) / len(big)))


def avg(agg, kernel_base, counter):
    """Mean over the full-size dispatches: the bench's set-up also launches the kernel on a handful of windows (the
    MARGIN_OLD solves that produce the priors), those stay out (anything below 80 % of the largest dispatch)."""
    vs = by_base(agg, kernel_base, counter)
    if not vs:
        return None
    top = max(vs)
    vs = [v for v in vs if v >= 0.8 * top] or vs
    return sum(vs) / len(vs)


def per_step(agg, kernel_base, counter):
    """Sum over ALL dispatches of a front-end kernel divided by the number of steps (pyr_down runs three times per step on
    three different level sizes: its dispatches are summed, not 3 x the largest)."""
    vs = by_base(agg, kernel_base, counter)
    steps = len(by_base(agg, ONCE_PER_STEP, counter))
    if not vs or not steps:
        return None, 0
    # a kernel that does not run in every step (lk_track: the first frame of a run has nothing to track) is averaged over
    # the steps it ran in, i.e. a steady-state step
    return sum(vs) / (steps if len(vs) >= steps else len(vs)), len(vs)


def main():
    if "--json" not in sys.argv:
        return text(sys.argv[1:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", required=True)
    ap.add_argument("--workload", required=True)
    ap.add_argument("--calib", nargs=2, required=True, metavar=("FETCH_DB", "WRITE_DB"))
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    a = ap.parse_args()
    GiB = float(1 << 30)
    cf, cw = load(a.calib[0]), load(a.calib[1])
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    calib = {}
    for kern, ctr, agg in (("read8_kernel", "FETCH_SIZE", cf), ("read16_kernel", "FETCH_SIZE", cf), ("read1_kernel", "FETCH_SIZE", cf),
                           ("write8_kernel", "WRITE_SIZE", cw)):
        v = avg(agg, kern, ctr)
        calib[kern] = {"reported_bytes": v * 1024.0 if v else None, "true_bytes": GiB, "factor": GiB / (v * 1024.0) if v else None}
    f8 = calib["read8_kernel"]["factor"] or 1.0
    f1 = calib["read1_kernel"]["factor"] or 1.0
    fw = calib["write8_kernel"]["factor"] or 1.0
    bf, bw = load(a.fetch), load(a.write)
    out = {"workload": a.workload,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB per dispatch, averaged); each counter "
                     "scaled by the factor measured on tools/microbench/pmc_calib (1 GiB streams, > Infinity Cache): 8 B/lane reads "
                     "for the f64 window kernel, 1 B/lane reads for the front-end, 8 B/lane writes; front-end kernels per steady-state "
                     "step (a kernel that skips the first frame is averaged over the steps it ran in)",
           "calibration": calib}
    wk = "vio_window_kernel"
    f, w = avg(bf, wk, "FETCH_SIZE"), avg(bw, wk, "WRITE_SIZE")
    if f is not None and w is not None:
        out["vio_window_kernel"] = {"fetch_reported_bytes": f * 1024, "write_reported_bytes": w * 1024,
                                    "bytes_per_launch": f * 1024 * f8 + w * 1024 * fw}
    fe_f = fe_w = 0.0
    per = {}
    for k in FRONTEND:
        (f, nf), (w, nw) = per_step(bf, k, "FETCH_SIZE"), per_step(bw, k, "WRITE_SIZE")
        if f is None or w is None:
            continue
        per[k] = {"fetch_reported_bytes_per_step": f * 1024, "write_reported_bytes_per_step": w * 1024, "dispatches": nf,
                  "bytes_per_step": f * 1024 * f1 + w * 1024 * fw}
        fe_f += f * 1024
        fe_w += w * 1024
    out["frontend_step"] = {"kernels": per, "steps": len(by_base(bf, ONCE_PER_STEP, "FETCH_SIZE")),
                            "bytes_per_launch": fe_f * f1 + fe_w * fw}
    json.dump(out, open(a.json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
