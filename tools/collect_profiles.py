"""Turn one tools/profile_round.sh output directory (gpurun_out/<tag>) into the tracked summaries under profiles/:
<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <round>_pmc_summary.csv (per-kernel medians of the PMC
passes), <round>_bench.json, <round>_all_configs.jsonl and traffic.json (HBM bytes per launch of the two big kernels).
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is taken 1:1.
usage: python tools/collect_profiles.py gpurun_out/r01c r01"""
import csv, json, os, shutil, statistics, sys
from collections import defaultdict

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def medians(path):
    per = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(out, f"{rnd}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "stats_configs", "configs_kernel_stats.csv")):       # all BASELINE configs in one traced run
    shutil.copy(os.path.join(src, "stats_configs", "configs_kernel_stats.csv"), os.path.join(out, f"{rnd}_kernel_stats_all_configs.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, f"{rnd}_bench.json"))
if os.path.exists(os.path.join(src, "configs.jsonl")):
    shutil.copy(os.path.join(src, "configs.jsonl"), os.path.join(out, f"{rnd}_all_configs.jsonl"))
for extra in ("tolerance_audit.json", "latency_b1.txt", "latency_breakdown.txt", "pytest_gpu.txt", "c4_breakdown.json", "c4_breakdown_full_window.json",
              "traffic_configs.json", "pmc_all_configs.csv"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(out, f"{rnd}_{extra}"))
fetch = medians(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"))
write = medians(os.path.join(src, "pmc_write", "write_counter_collection.csv"))
sq = medians(os.path.join(src, "pmc_sq", "sq_counter_collection.csv"))
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT",
         "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
# the big launches only: take the upper half of each kernel's samples (warm-up and the 8-clip check are tiny launches)
def big_median(v):
    v = sorted(v)
    return statistics.median(v[len(v) // 2:]) if v else 0.0
with open(os.path.join(out, f"{rnd}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_sampled", "FETCH_SIZE_KB_median", "WRITE_SIZE_KB_median"] + names)
    for k in sorted(set(fetch) | set(write) | set(sq)):
        w.writerow([k, len(fetch[k].get("FETCH_SIZE", [])), big_median(fetch[k].get("FETCH_SIZE", [])),
                    big_median(write[k].get("WRITE_SIZE", []))] + [big_median(sq[k].get(n, [])) for n in names])
traffic = {"_note": "HBM bytes per launch at batch 4096 from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, median "
                    "of the full-size launches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                    "coalesced reads); WRITE_SIZE taken 1:1."}
plan = {"cnn_trunk_b_kernel": "trunk_x3:conv1+pool+conv2+pool", "cnn_trunk_h2_kernel": "trunk_x3:conv1+pool+conv2+pool", "cnn_trunk_kernel": "trunk:conv1+pool+conv2+pool",
        "fe_stft_mel_db_kernel": "frontend:fe_stft_mel_db_kernel", "fe2_wave_kernel": "frontend:fe_stft_mel_db_kernel"}
for prefix, label in plan.items():
    k = next((n for n in fetch if n.startswith(prefix + "<") or n == prefix), None)
    if k is not None and k in write:
        fk, wk = big_median(fetch[k]["FETCH_SIZE"]), big_median(write[k]["WRITE_SIZE"])
        traffic[label] = {"batch": 4096, "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "fetch_size_kb": fk, "write_size_kb": wk}
        s = sq.get(k, {})
        if s:
            traffic[label]["sq_insts_valu"] = int(big_median(s.get("SQ_INSTS_VALU", [])))      # bench.py: stft_stage.valu_issue_frac
            traffic[label]["sq_insts_mfma"] = int(big_median(s.get("SQ_INSTS_MFMA", [])))
            busy, act = big_median(s.get("SQ_BUSY_CYCLES", [])), big_median(s.get("SQ_LDS_IDX_ACTIVE", []))
            print(f"{label}: FETCHx2 {2*fk/1024:.1f} MB WRITE {wk/1024:.1f} MB | INSTS_VALU {big_median(s.get('SQ_INSTS_VALU', [])):.3g} "
                  f"INSTS_MFMA {big_median(s.get('SQ_INSTS_MFMA', [])):.3g} MFMA_BUSY {big_median(s.get('SQ_VALU_MFMA_BUSY_CYCLES', [])):.3g} "
                  f"LDS_ACTIVE {act:.3g} LDS_CONFLICT {big_median(s.get('SQ_LDS_BANK_CONFLICT', [])):.3g} SQ_BUSY {busy:.3g}")
tc = os.path.join(src, "traffic_configs.json")
if os.path.exists(tc):      # every kernel of the non-headline configs (tools/traffic_configs.sh), at the batch of its config
    traffic["configs_per_kernel"] = {"_note": "C3 (B = 8192, fp32 and bf16 activations), C5 (B = 2048), C4 (1024 streams), e2e / gru (B = 4096): "
                                              "FETCH_SIZE x 2 + WRITE_SIZE per launch, rocprofv3 --pmc, one counter per pass",
                                     **json.load(open(tc))}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(out)))
