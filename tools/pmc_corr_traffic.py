#!/usr/bin/env python
"""profiles/pmc/rNN_corr_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass,
kernel-filtered to the correlation launch) over THE SAME bench.py command line (tools/r05_pmc_corr.sh).
The launches are segmented by dispatch order, counted from the end of the run:
    ... | warm-up W + timed K | I instrumented | 4 re-entry | L compact (every factor live, unit spacing) | L wrapped (every factor live)
and the figures are means over the timed segment (and the compact / wrapped ones), i.e. at the benchmarked state's own
factor count -- not extrapolated from a shorter run.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE x 2 on gfx950
(MI355X_MICROARCH.md: wide coalesced reads are tallied at half their bytes); factors per launch = WRITE_SIZE / 1792 B
(every factor writes one row of 896 halfs)."""
import csv, json, sys


def values(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "corr_mfma" in (r.get("Kernel_Name") or "")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fetch_csv, write_csv, K, W, L, I, out, cmd = (sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                                                  int(sys.argv[6]), sys.argv[7], sys.argv[8])
    row_bytes = float(sys.argv[9]) if len(sys.argv) > 9 else 1792.0          # one output row per factor: 896 halfs (fp32: 3584)
    kname = sys.argv[10] if len(sys.argv) > 10 else "corr_mfma_kernel<_Float16, true>"
    f, w = values(fetch_csv, "FETCH_SIZE"), values(write_csv, "WRITE_SIZE")
    assert len(f) == len(w) and len(f) >= K + W + I + 4 + 2 * L, (len(f), len(w))
    seg = {"live_wrapped": slice(len(f) - L + 2, len(f)), "live_compact": slice(len(f) - 2 * L + 2, len(f) - L),
           "timed": slice(len(f) - 2 * L - 4 - I - K, len(f) - 2 * L - 4 - I)}
    res = {"kernel": kname, "command": cmd,
           "source": "tools/r05_pmc_corr.sh -> tools/pmc_corr_traffic.py; counters collected for the correlation launches only "
                     "(--kernel-include-regex), one counter per pass, the tracker's cross-stream signal words replaced by "
                     "events under counter collection (rampvo_amd/Ramp_vo.py::_kernels_are_serialised)",
           "fetch_correction": 2.0, "dispatches": len(f)}
    for name, s in seg.items():
        fs, ws = f[s], w[s]
        E = [x * 1024.0 / row_bytes for x in ws]
        per = [(2.0 * a + b) * 1024.0 / e for a, b, e in zip(fs, ws, E)]
        res[name] = {"launches": len(fs), "fetch_size_kb_per_launch": round(sum(fs) / len(fs), 1),
                     "write_size_kb_per_launch": round(sum(ws) / len(ws), 1), "edges_per_launch": int(round(sum(E) / len(E))),
                     "bytes_per_launch": int(sum((2.0 * a + b) * 1024.0 for a, b in zip(fs, ws)) / len(fs)),
                     "bytes_per_edge": round(sum(per) / len(per), 1)}
    res["edges_per_launch"] = res["timed"]["edges_per_launch"]         # (the keys bench.py reads)
    res["bytes_per_edge"] = res["timed"]["bytes_per_edge"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


main()
