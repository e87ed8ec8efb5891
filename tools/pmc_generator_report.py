"""Joins tools/pmc_generator.py's launch log with the three rocprofv3 passes (by launch order within a kernel family) and writes
<tag>_pmc_generator.json (conv_pair_kernel, conv_gemm2win_kernel, the other convolutions) and <tag>_pmc_synth.json."""
import csv, glob, json, os, sys
O, tag = sys.argv[1], sys.argv[2]
log = json.load(open(os.path.join(O, "launch_log.json")))["launches"]
FAM = {"conv_pair": "conv_pair_kernel", "conv_gemm2win": "conv_gemm2win_kernel", "additive_synth": "additive_synth_kernel"}


def fam_of(name):
    for f, key in FAM.items():
        if key in name:
            return f
    return "conv_gemm_other" if "conv_gemm" in name else None


def ordered(path, value):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Start_Timestamp") or r.get("Dispatch_Id") or 0))
    out = {}
    for r in rows:
        f = fam_of(r["Kernel_Name"])
        if f:
            out.setdefault(f, []).append((value(r), r["Kernel_Name"].split("(")[0][-70:]))
    return out


kt = ordered(glob.glob(f"{O}/kt/*/*kernel_trace.csv")[0], lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def pmc(d, counter):
    f = glob.glob(f"{O}/{d}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = {}
    for r in rows:
        fm = fam_of(r["Kernel_Name"])
        if fm:
            out.setdefault(fm, []).append(float(r["Counter_Value"]))
    return out


fetch, write = pmc("pf", "FETCH_SIZE"), pmc("pw", "WRITE_SIZE")
per = {}
for fam in sorted({l["family"] for l in log}):
    mine = [l for l in log if l["family"] == fam]
    n = len(mine)
    dur, fe, wr = kt.get(fam, [])[-n:], fetch.get(fam, [])[-n:], write.get(fam, [])[-n:]      # the logged forward is the LAST one of the run
    assert len(dur) == len(fe) == len(wr) == n, (fam, n, len(dur), len(fe), len(wr))
    groups = {}
    for l, (us, kname), f_kib, w_kib in zip(mine, dur, fe, wr):
        key = (l["kernel"], l["m"], l["n"], l["cin"], l["taps"], l["resid"], l["accumulate"])
        g = groups.setdefault(key, dict(kernel=kname, tile=l["kernel"], rows=l["m"], cout=l["n"], cin=l["cin"], taps=l["taps"], residual=l["resid"],
                                        accumulate=l["accumulate"], launches=0, us=0.0, fetch_kib=0.0, write_kib=0.0, alg_bytes=l["alg_bytes"]))
        g["launches"] += 1; g["us"] += us; g["fetch_kib"] += f_kib; g["write_kib"] += w_kib
    rows = []
    for g in groups.values():
        k = g["launches"]
        hbm = (2 * g["fetch_kib"] + g["write_kib"]) * 1024 / k            # FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B), KiB units
        us = g["us"] / k
        rows.append(dict(kernel=g["kernel"], tile=g["tile"], rows=g["rows"], cout=g["cout"], cin=g["cin"], taps=g["taps"], residual=g["residual"],
                         accumulate=g["accumulate"], launches=k, avg_us=round(us, 2), fetch_size_kib=round(g["fetch_kib"] / k, 1),
                         write_size_kib=round(g["write_kib"] / k, 1), hbm_bytes_per_launch=int(hbm), algorithmic_bytes=g["alg_bytes"],
                         traffic_ratio=round(hbm / g["alg_bytes"], 3), hbm_gb_s=round(hbm / us / 1e3, 1),
                         algorithmic_gb_s=round(g["alg_bytes"] / us / 1e3, 1), frac_of_6300_gb_s=round(hbm / us / 1e3 / 6300, 3)))
    rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
    per[fam] = rows
note = ("one eager generator forward at 1500 frames, ResBlock branches in series; FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes, "
        "durations from a third pass with --kernel-trace only; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB units, gfx950 correction); "
        "algorithmic bytes = 4 x (input rows x cin + output rows x cout + weights) (+ residual / accumulate operand); the fused pair's inner "
        "activation never leaves the chip; 6300 GB/s = the achievable HBM rate of MI355X_MICROARCH.md")
json.dump(dict(note=note, conv_pair_kernel=per.get("conv_pair", []), conv_gemm2win_kernel=per.get("conv_gemm2win", []),
               other_convolutions=per.get("conv_gemm_other", [])), open(os.path.join(O, f"{tag}_pmc_generator.json"), "w"), indent=1)
json.dump(dict(note=note, additive_synth_kernel=per.get("additive_synth", [])), open(os.path.join(O, f"{tag}_pmc_synth.json"), "w"), indent=1)
for fam, rows in per.items():
    t = sum(r["avg_us"] * r["launches"] for r in rows)
    hb = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows); ab = sum(r["algorithmic_bytes"] * r["launches"] for r in rows)
    print(f"{fam:18s} {sum(r['launches'] for r in rows):3d} launches {t:8.1f} us  HBM {hb / 1e6:8.1f} MB vs algorithmic {ab / 1e6:8.1f} MB (x{hb / ab:.2f})  {hb / t / 1e3:7.1f} GB/s")
