"""Add derived metrics to a pmc_summary.py JSON (counters summed over the 8 XCDs by rocprofv3).
usage: pmc_derive.py in.json out.json [key=value ...]   (extra key=value pairs are recorded verbatim)"""
import sys, json
p = json.load(open(sys.argv[1]))
dur = [v for k, v in p.items() if k.startswith("dur_ms_")]
d_main = next(v for k, v in p.items() if k.startswith("dur_ms_GRBM"))
clk = p["GRBM_GUI_ACTIVE"] / 8 / (d_main * 1e-3)  # Hz
p["clock_GHz"] = clk / 1e9
p["MfmaUtil_pct"] = 100.0 * p["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / 8 * 1024)  # 1024 SIMDs
d_mf = next(v for k, v in p.items() if k.startswith("dur_ms_SQ_ACTIVE_INST_LDS"))
p["mfma_flops_TF_issued"] = p["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (d_mf * 1e-3) / 1e12
if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in p:  # (the bf16-pipe network: every float32 product is six of these)
    p["mfma_bf16_flops_TF_issued"] = p["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (d_mf * 1e-3) / 1e12
if p.get("SQ_LDS_IDX_ACTIVE"):
    p["lds_conflict_frac"] = p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"]
p["wave_cycles_parked_frac"] = p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"]
p["wave_cycles_issue_stall_frac"] = p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"]
p["wave_cycles_active_frac"] = p["SQ_ACTIVE_INST_ANY"] / p["SQ_WAVE_CYCLES"]
p["hbm_read_GBps_raw"] = p["FETCH_SIZE"] * 1024 / (p["dur_ms_FETCH_SIZE"] * 1e-3) / 1e9   # FETCH_SIZE / WRITE_SIZE are in KiB
p["hbm_write_GBps"] = p["WRITE_SIZE"] * 1024 / (p["dur_ms_WRITE_SIZE"] * 1e-3) / 1e9
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    try:
        p[k] = v if k in ("commit", "passes_dir", "kernel", "command") else float(v)
    except ValueError:
        p[k] = v
# per-evaluation figures: every pass is its own run of the same command; its bench line says how many evaluations its timed
# dispatch performed (the counts differ by < 1 % between passes: the visit pools are drawn in arrival order)
if "passes_dir" in p:
    import glob, os
    ev = {}
    for log in sorted(glob.glob(os.path.join(str(p["passes_dir"]), "pass*.log"))):
        lines = [l for l in open(log) if l.startswith('{"metric"')]
        if lines:
            ev[os.path.basename(log)[:-4]] = json.loads(lines[-1]).get("evals_timed_rank0")
    p["evals_per_pass"] = ev
    vals = [v for v in ev.values() if v]
    if vals:
        p["evals_in_dispatch"] = sum(vals) / len(vals)
        e2, e3, e4 = ev.get("pass2") or p["evals_in_dispatch"], ev.get("pass3") or p["evals_in_dispatch"], ev.get("pass4") or p["evals_in_dispatch"]
        p["mfma_bf16_instructions_per_eval"] = p.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) * 512 / 16384 / e2   # K = 32, 16 x 16 tiles
        p["valu_instructions_per_eval"] = p["SQ_INSTS_VALU"] / (ev.get("pass1") or p["evals_in_dispatch"])
        p["hbm_bytes_per_eval"] = p["FETCH_SIZE"] * 1024 / e3 + p["WRITE_SIZE"] * 1024 / e4
        p["lds_busy_frac"] = p["SQ_LDS_IDX_ACTIVE"] / (p["GRBM_GUI_ACTIVE"] / 8 * 256)   # LDS-array cycles / (cycles x 256 CUs)
        p["valu_issue_frac"] = p["SQ_ACTIVE_INST_VALU"] * 4 / (p["GRBM_GUI_ACTIVE"] / 8 * 1024)  # quad-cycles -> cycles / (cycles x 1024 SIMDs)
json.dump(p, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: p[k] for k in p if not k.startswith("SQ_") and not k.startswith("dur_")}, indent=1))
