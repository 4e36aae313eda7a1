#!/usr/bin/env python3
"""HBM-side traffic per launch of each kernel class, from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_run.sh.
usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag>  > profiles/pmc_traffic.json
Units: rocprofv3 reports both counters in KiB.  Correction: the axpby calibration passes (tools/pmc_calib.py, same
4-byte-per-lane coalesced access as the library's kernels, known byte count) give counter/true-bytes factors; the
training-step counters are divided by them (MI355X_MICROARCH.md: gfx950's FETCH_SIZE under-reports coalesced reads)."""
import collections, csv, glob, json, sys
d = sys.argv[1]
CLASSES = [("conv1d_fwd_kernel", "conv_mfma"), ("gated_block_fwd", "block_fwd"), ("gated_block_dgrad", "block_dgrad"),
           ("gated_block_wgrad", "block_wgrad"), ("conv1d_wgrad_kernel", "wgrad_mfma"), ("quantize_fwd_kernel", "quantize_fwd"),
           ("conv1d_cout1_kernel", "conv_cout1"), ("conv_split_kernel", "conv_split"), ("conv_wgrad_split_kernel", "conv_wgrad_split")]


def read(sub, counter, by_grid=False):
    """kernel name [@grid<workgroups>] -> (sum of counter, launches).  by_grid buckets the launches of one kernel by
    their grid size, so that a record describes ONE launch shape (the quantizer runs at the training batch - 128
    workgroups - and at the config-5 batch - 1024 - in the same profile)."""
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            if by_grid:
                wg = int(r.get("Grid_Size", 0)) // max(int(r.get("Workgroup_Size", 1)), 1)
                k = f"{k}@grid{wg}"
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt


def calib(sub, counter, true_bytes):
    tot, cnt = read(sub, counter)
    k = [n for n in tot if "axpby" in n]
    if not k:
        return None
    return (tot[k[0]] / cnt[k[0]]) * 1024.0 / true_bytes


n = 128 * 1024 * 1024
cf, cw = calib("calib_fetch", "FETCH_SIZE", 8.0 * n), calib("calib_write", "WRITE_SIZE", 4.0 * n)
ft, fc = read("pass3", "FETCH_SIZE")
wt, wc = read("pass4", "WRITE_SIZE")
out = {"_calibration": {"fetch_counter_per_true_byte": cf, "write_counter_per_true_byte": cw,
                        "how": "nsc_axpby over 128 Mi floats (1 GiB read, 512 MiB written per launch), 4 B/lane coalesced"}}
for pat, tag in CLASSES:
    fk = [k for k in ft if pat in k]; wk = [k for k in wt if pat in k]
    nf, nw = sum(fc[k] for k in fk), sum(wc[k] for k in wk)
    if not nf or not nw:
        continue
    fb = sum(ft[k] for k in fk) * 1024.0 / nf / (cf or 1.0)
    wb = sum(wt[k] for k in wk) * 1024.0 / nw / (cw or 1.0)
    out[tag] = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "launches_sampled": int(nf),
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) in separate passes over bench.py --no-overlap --no-infer "
                          "(training-step launches only), divided by the axpby calibration factors; averaged over the launches "
                          "of the kernel family"}
# per launch SHAPE for the kernels that run at more than one grid size in the profiled command
ftg, fcg = read("pass3", "FETCH_SIZE", True)
wtg, wcg = read("pass4", "WRITE_SIZE", True)
# (the op-surface quantizer forward at inference batch sizes is quantize_fwd32_wave_kernel: bench.py's roofline_quantizer record)
for pat, tag in (("quantize_fwd_kernel", "quantize_fwd"), ("quantize_fwd32_wave_kernel", "quantize_fwd_wave"),
                 ("gated_block_dgrad2", "block_dgrad"), ("gated_block_fwd2", "block_fwd")):
    grids = sorted({k.split("@grid")[1] for k in ftg if pat in k})
    for g in grids:
        fk = [k for k in ftg if pat in k and k.endswith("@grid" + g)]
        wk = [k for k in wtg if pat in k and k.endswith("@grid" + g)]
        nf, nw = sum(fcg[k] for k in fk), sum(wcg[k] for k in wk)
        if nf and nw:
            out[f"{tag}@grid{g}"] = {"fetch_bytes_per_launch": round(sum(ftg[k] for k in fk) * 1024.0 / nf / (cf or 1.0)),
                                     "write_bytes_per_launch": round(sum(wtg[k] for k in wk) * 1024.0 / nw / (cw or 1.0)),
                                     "launches_sampled": int(nf), "workgroups": int(g)}
json.dump(out, sys.stdout, indent=1)
