"""profiles/traffic.json entries from the counters of RESIDENT launches (scripts/gpu_prof_resident.sh -> counters_<tag>.json, committed
as profiles/rNN/counters_<tag>_resident.json): per 64-frame step of a 300-step launch. HBM bytes = 2 x FETCH_SIZE (KB; the gfx950
correction of MI355X_MICROARCH.md's HBM section) + WRITE_SIZE (KB), separate --pmc passes.
    python scripts/traffic_from_resident.py r06"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
# (tag, kernel function as nam_hip_batch_kernel_name reports it, model, streams, waves per SIMD, SIMDs that hold a wave,
#  measured issue cycles per non-matrix vector instruction at that occupancy + where it was measured)
RUNS = [
    ("c4_wn_reg", "nam_wn_reg_kernel", "wavenet_a2_max", 512, 1, 1024, 6.79,
     "profiles/r05/valu_rate_microbench.txt: ONE wave per SIMD, v_fma_f32 + v_mul_f32 interleaved 6.79 cycles per instruction of the wave "
     "(plain v_fma_f32 8.61, v_pk_fma_f32 9.58: the lowest of the one-wave rows = the floor). Config 4 runs two waves per stream = one per SIMD."),
    ("c5_wn_reg", "nam_wn_reg_kernel", "slimmable_wavenet", 768, 2, 1024, 3.40,
     "profiles/r05/valu_rate_microbench.txt: TWO waves per SIMD, v_fma_f32 + v_mul_f32 interleaved 3.40 cycles per SIMD-instruction (v_pk_fma_f32 5.05: "
     "the lowest of the two-wave rows = the floor). Config 5 runs the dense form (round 6): two waves per stream, 1,536 waves = one or two on "
     "every one of the 1,024 SIMDs."),
    ("c2_q", "nam_a1_q_kernel", "wavenet_a1_standard", 256, 4, 1024, 3.84,
     "tools/src/valu_rate.hip, four waves per SIMD, the stage bodies' own mix next to the kernel's fp32 matrix instructions (profiles/r05/valu_rate_microbench.txt)"),
    ("a2_kq", "nam_kq_kernel", "A2", 256, 4, 1024, 4.06,
     "tools/src/valu_rate.hip, four waves per SIMD, 4x4x1 matrix instructions + plain vector instructions (profiles/r05/valu_rate_microbench.txt)"),
]
path = os.path.join(ROOT, "profiles", "traffic.json")
doc = json.load(open(path))
for tag, kernel, model, streams, wps, simds, cpi, cpi_note in RUNS:
    src = os.path.join("profiles", rnd, f"counters_{tag}_resident.json")
    if not os.path.exists(os.path.join(ROOT, src)):
        print("missing", src, file=sys.stderr)
        continue
    c = json.load(open(os.path.join(ROOT, src)))
    s = c["per_step"]
    e = {
        "kernel": kernel, "model": model, "streams": streams, "block": 64, "launch": "block",
        "hbm_bytes_per_launch": int((2 * s.get("FETCH_SIZE", 0.0) + s.get("WRITE_SIZE", 0.0)) * 1024),
        "kernel_cycles": s.get("GRBM_GUI_ACTIVE"),
        "lds_idx_active_cycles": s.get("SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict_cycles": s.get("SQ_LDS_BANK_CONFLICT"),
        "insts_per_launch": {"VALU": s.get("SQ_INSTS_VALU"), "SALU": s.get("SQ_INSTS_SALU"), "SMEM": s.get("SQ_INSTS_SMEM"),
                             "LDS": s.get("SQ_INSTS_LDS"), "VMEM_RD": s.get("SQ_INSTS_VMEM_RD"), "VMEM_WR": s.get("SQ_INSTS_VMEM_WR")},
        "mfma_mops_f32": s.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), "mfma_insts": s.get("SQ_INSTS_MFMA", 0.0),
        "mfma_busy_cycles": s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
        "rocprof_avg_launch_us": round(c["ns_per_step"] / 1e3, 3),
        "waves_per_simd": wps, "simds_busy": simds,
        "issue_cycles_per_valu_inst": cpi, "issue_note": cpi_note,
        "source": src,
        "note": f"{src}: 300-step resident launches behind the bench's spin-up (settled clocks), per 64-frame step; rocprofv3 --kernel-trace --stats "
                f"+ separate --pmc passes (scripts/gpu_prof_resident.sh); HBM = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
        "lds_note": "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT of the resident launch / 300",
    }
    doc["entries"] = [x for x in doc["entries"] if (x["kernel"], x.get("model"), x["streams"], x["launch"]) != (kernel, model, streams, "block")]
    doc["entries"].append(e)
    print(f"{tag}: {e['rocprof_avg_launch_us']} us per step, VALU {e['insts_per_launch']['VALU']:.0f}, HBM {e['hbm_bytes_per_launch']} B")
json.dump(doc, open(path, "w"), indent=1)
