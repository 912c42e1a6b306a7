"""Build profiles/traffic.json from the round's committed rocprofv3 summaries (profiles/rNN/rocprofv3_summary_*.txt,
written by scripts/gpu_profile_config.sh + scripts/summarize_prof.py). One entry per (kernel function, model, streams,
block, launch); bench.py looks its run up by exactly that key.

HBM bytes per launch = 2 x FETCH_SIZE (KB; gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md, HBM
section) + WRITE_SIZE (KB), each from its own --pmc pass."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [  # (summary tag, kernel function, model, streams, launch)
    ("r02", "c2_p2", "nam_a1_p2_kernel", "wavenet_a1_standard", 256, "block"),
    ("r02", "c3_lstm_row", "nam_lstm_row_kernel", "lstm", 1024, "block"),
    ("r02", "c4_wn_reg", "nam_wn_reg_kernel", "wavenet_a2_max", 512, "block"),
    ("r02", "c5_wn_reg", "nam_wn_reg_kernel", "slimmable_wavenet", 768, "block"),
    ("r02", "c4_generic", "nam_generic_kernel", "wavenet_a2_max", 512, "block"),
    ("r02", "c2_valu", "nam_a1_kernel", "wavenet_a1_standard", 256, "block"),
    # round 3: config 4 again (conv-by-tap / shape-set changes since round 2; the container model is an exact ahead-of-time shape set)
    ("r03", "c4_wn_reg", "nam_wn_reg_kernel", "wavenet_a2_max", 512, "block"),
]


def counters(path, kernel):
    """{counter: per-dispatch average} of the kernel's block-launch instantiation (the one with the most dispatches)."""
    best, out = 0, {}
    for line in open(path):
        if kernel not in line or "dispatches=" not in line:
            continue
        n = int(re.search(r"dispatches=(\d+)", line).group(1))
        vals = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.e+]+)", line) if k != "dispatches"}
        if n >= best:
            if n > best:
                out = {}
            best = n
            out.update(vals)
    avg_ns = None
    for line in open(path):
        if kernel in line and "avg_ns=" in line:
            c = int(re.search(r"calls=(\d+)", line).group(1))
            if c >= best * 0.9:
                avg_ns = float(re.search(r"avg_ns=([0-9.]+)", line).group(1))
    return best, out, avg_ns


entries = [json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"][0]]  # round 1: a1_mfma (kept)
entries = [e for e in entries if e.get("kernel") == "a1_mfma"]
for e in entries:
    e["kernel"] = "nam_a1_mfma_kernel"
    e.setdefault("model", "wavenet_a1_standard")
for rnd, tag, kernel, model, streams, launch in RUNS:
    path = os.path.join(ROOT, "profiles", rnd, f"rocprofv3_summary_{tag}.txt")
    n, c, avg_ns = counters(path, kernel)
    if not n:
        print("no counters for", tag, file=sys.stderr)
        continue
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    entries = [e for e in entries if (e["kernel"], e["model"], e["streams"], e["launch"]) != (kernel, model, streams, launch)]
    entries.append({
        "kernel": kernel, "model": model, "streams": streams, "block": 64, "launch": launch,
        "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
        "kernel_cycles": c.get("GRBM_GUI_ACTIVE"),
        "lds_idx_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
        "insts_per_launch": {k: c.get("SQ_INSTS_" + k) for k in ("VALU", "SALU", "SMEM", "LDS", "VMEM_RD", "VMEM_WR")},
        "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
        "rocprof_avg_launch_us": None if avg_ns is None else round(avg_ns / 1e3, 2),
        "note": f"profiles/{rnd}/rocprofv3_summary_{tag}.txt ({n} dispatches, one launch per 64-frame step): FETCH_SIZE {fetch_kb:,.0f} KB x2 "
                f"(gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE {write_kb:,.0f} KB, separate "
                "--pmc passes; the persistent block mode runs the same kernel body per command",
        "lds_note": "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE, per launch (kernel-trace only)",
    })
# round 3: nam_a1_p4_kernel pipelines CONSECUTIVE buffers, so a single 64-frame launch never runs it: its counters come
# from a resident launch (one dispatch = 300 steps; scripts/resident_counters.py keeps the dispatches apart) divided by 300.
# The persistent block mode's session launch runs the same loop body per command.
RESIDENT = [("r03", "counters_c2_p4_resident.json", "nam_a1_p4_kernel", "wavenet_a1_standard", 256, 300),
            ("r03", "counters_a2_kp_resident.json", "nam_kp_kernel", "A2", 256, 300),
            ("r03", "counters_a2_kq_resident.json", "nam_kq_kernel", "A2", 256, 300),
            # nam_wn_reg_kernel as the sessions run it (two wavefronts per stream for config 4): the history never leaves LDS
            ("r03", "counters_c4_wn_reg2_resident.json", "nam_wn_reg_kernel", "wavenet_a2_max", 512, 300),
            ("r03", "counters_c5_wn_reg_resident.json", "nam_wn_reg_kernel", "slimmable_wavenet", 768, 300)]
for rnd, name, kernel, model, streams, steps in RESIDENT:
    path = os.path.join(ROOT, "profiles", rnd, name)
    if not os.path.exists(path):
        continue
    d = max(json.load(open(path))["dispatches"], key=lambda x: x["ns"])  # the K-step launch
    per = lambda k: None if d.get(k) is None else d[k] / steps
    entries = [e for e in entries if not (e["kernel"] == kernel and e["model"] == model and e["streams"] == streams)]
    entries.append({
        "kernel": kernel, "model": model, "streams": streams, "block": 64, "launch": "block",
        "hbm_bytes_per_launch": int((2 * per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024),
        "kernel_cycles": per("GRBM_GUI_ACTIVE"),
        "lds_idx_active_cycles": per("SQ_LDS_IDX_ACTIVE"),
        "lds_bank_conflict_cycles": per("SQ_LDS_BANK_CONFLICT"),
        "insts_per_launch": {k: per("SQ_INSTS_" + k) for k in ("VALU", "SALU", "SMEM", "LDS", "VMEM_RD", "VMEM_WR")},
        "mfma_mops_f32": per("SQ_INSTS_VALU_MFMA_MOPS_F32"),
        "rocprof_avg_launch_us": round(d["ns"] / steps / 1e3, 3),
        "note": f"profiles/{rnd}/{name}: ONE resident launch of {steps} steps (rocprofv3 --kernel-trace: {d['ns'] / 1e3:,.0f} us), every counter "
                f"divided by {steps} -> per 64-frame step: FETCH_SIZE {per('FETCH_SIZE'):,.0f} KB x2 (gfx950 reports half of wide coalesced "
                f"reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE {per('WRITE_SIZE'):,.0f} KB, separate --pmc passes; the kernel pipelines "
                "consecutive buffers, so a lone 64-frame launch never runs it; the persistent block mode runs the same loop body per command",
        "lds_note": "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE of the resident launch / steps",
    })
# the issue-port terms of bench.py's floor (fp32 MFMA pipe cycles, MFMA instructions) for the round-3 entries, from their own
# counters: nam_a1_p4_kernel issues 16x16x4 instructions only (4 "MOPS" = 32 pipe cycles each), nam_kq_kernel 4x4x1 only
# (761,856 per step, counted: profiles/r03/counters_kq_resident.txt; 8 pipe cycles each)
for e in entries:
    if e["kernel"] == "nam_a1_p4_kernel" and e.get("mfma_mops_f32"):
        e["mfma_insts"] = e["mfma_mops_f32"] / 4.0
        e["mfma_busy_cycles"] = e["mfma_insts"] * 32.0
    if e["kernel"] == "nam_kq_kernel" and e["model"] == "A2":
        e["mfma_insts"] = 761856.0
        e["mfma_busy_cycles"] = 761856.0 * 8.0
# round 4: scripts/gpu_prof_resident.sh writes counters_<tag>.json = {kernel, ns_per_step, per_step: {counter: value}} — a 300-step
# resident launch of a later repetition (settled clocks), every counter divided by 300
RESIDENT_R4 = [("r04", "counters_c2_q_resident.json", "nam_a1_q_kernel", "wavenet_a1_standard", 256),
               ("r04", "counters_a2_kq_resident.json", "nam_kq_kernel", "A2", 256),
               ("r04", "counters_c3_lstm_row_resident.json", "nam_lstm_row_kernel", "lstm", 1024)]
for rnd, name, kernel, model, streams in RESIDENT_R4:
    path = os.path.join(ROOT, "profiles", rnd, name)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    c = d["per_step"]
    entries = [e for e in entries if not (e["kernel"] == kernel and e["model"] == model and e["streams"] == streams)]
    entries.append({
        "kernel": kernel, "model": model, "streams": streams, "block": 64, "launch": "block",
        "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
        "kernel_cycles": c.get("GRBM_GUI_ACTIVE"),
        "lds_idx_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
        "insts_per_launch": {k: c.get("SQ_INSTS_" + k) for k in ("VALU", "SALU", "SMEM", "LDS", "VMEM_RD", "VMEM_WR")},
        "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
        "mfma_insts": c.get("SQ_INSTS_MFMA"),
        "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"),
        "rocprof_avg_launch_us": round(d["ns_per_step"] / 1e3, 3),
        "note": f"profiles/{rnd}/{name}: 300-step resident launches behind the bench's own spin-up (median {d['ns_per_step'] / 1e3:.3f} us per step), "
                f"every counter divided by 300: FETCH_SIZE {c['FETCH_SIZE']:,.0f} KB x2 (gfx950 reports half of wide coalesced reads, "
                f"MI355X_MICROARCH.md HBM section) + WRITE_SIZE {c['WRITE_SIZE']:,.0f} KB, separate --pmc passes (scripts/gpu_prof_resident.sh); "
                "the kernel pipelines consecutive buffers, so a lone 64-frame launch never runs it; a session's launch runs the same loop body per command",
        "lds_note": "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT of the resident launch / steps",
    })
json.dump({"entries": entries}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for e in entries:
    print(e["kernel"], e["model"], e["streams"], e["hbm_bytes_per_launch"], e.get("rocprof_avg_launch_us"), e.get("insts_per_launch"))
