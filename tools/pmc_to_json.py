"""Turn the summary of tools/pmc_pass.sh (summary.json + the kernel traces of its passes) into profiles/rNN_pmc.json: the two
entries bench.py reads for the headline kernel (4K x 32 and 8K x 8 launches) and, under "other_kernels", one row per other
rwh:: kernel of the bench with the busy fractions that name its limiter.   usage: python tools/pmc_to_json.py <pmc dir> <out json>"""
import csv, glob, json, os, statistics, sys
root, out = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(root, "summary.json")))
KERNEL = "void rwh::warp_rgb8_fast8<unsigned char, 6>(rwh::FastArgs)"
def kernel_us(name, grid):
    f = glob.glob(os.path.join(root, "pass1", "**", "*kernel_trace.csv"), recursive=True)[0]
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
         if r["Kernel_Name"] == name and r.get("Grid_Size", r.get("Grid_Size_X")) == str(grid)]
    return statistics.mean(v[len(v) // 2:]) if v else float("nan")
def facts(name, grid):
    c = summ["%s|%d" % (name, grid)]
    cyc = c["SQ_BUSY_CYCLES"] / 32                      # 32 shader engines
    us = kernel_us(name, grid)
    if us != us:
        raise KeyError("no launch of this kernel / grid in the trace of pass 1")
    e = {"waves_per_launch": round(c["SQ_WAVES"]), "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1),
         "salu_insts_per_wave": round(c["SQ_INSTS_SALU"] / c["SQ_WAVES"], 1), "kernel_cycles": round(cyc), "sclk_hz": round(cyc / (us * 1e-6)),
         "kernel_us_in_pmc_pass": round(us, 1), "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3),
         "cycles_per_valu_inst": round(c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"], 3), "ta_busy_frac": round(c["TA_BUSY_avr"] / cyc, 3),
         "lds_busy_frac": round(c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 3),
         "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3),
         "vmem_rd_insts_per_wave": round(c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], 2), "vmem_wr_insts_per_wave": round(c["SQ_INSTS_VMEM_WR"] / c["SQ_WAVES"], 2),
         "FETCH_SIZE_KB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KB_raw": c["WRITE_SIZE"],
         "hbm_read_bytes_per_launch": round(c["FETCH_SIZE"] * 2048), "hbm_write_bytes_per_launch": round(c["WRITE_SIZE"] * 1024)}
    if "TCC_BUSY_avr" in c:      # round 4: the L2's own busy fraction, its requests, what the CUs' L1s waited for
        e.update({"tcc_busy_frac": round(c["TCC_BUSY_avr"] / cyc, 3), "tcc_requests": round(c.get("TCC_REQ_sum", 0)),
                  "l2_read_requests": round(c.get("TCP_TCC_READ_REQ_sum", 0)), "l2_write_requests": round(c.get("TCP_TCC_WRITE_REQ_sum", 0)),
                  "tcp_pending_stall_frac": round(c.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / cyc, 3)})
    e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    return e
doc = {}
for key, grid, src, frames, alg in (("rwh::warp_rgb8_fast8<unsigned char, 6>", 31211520, [3840, 2160], 32, 1530430848),
                                    ("rwh::warp_rgb8_fast8<unsigned char, 6> @ 8 x 7680x4320", 28483584, [7680, 4320], 8, 1476410496)):
    if "%s|%d" % (KERNEL, grid) not in summ:
        continue
    e = {"src": src, "frames": frames}
    e.update(facts(KERNEL, grid))
    e["algorithmic_bytes_per_launch"] = alg
    e["limiter"] = ("VALU busy %d %% (%d VALU instructions per 512-pixel wave at %.2f cycles each), L1s waiting on L2 / HBM %d %% of the cycles, texture-address path busy %d %%, LDS busy %d %% "
                    "(%d %% of it bank conflicts); HBM traffic = %.4f x algorithmic; the access pattern alone (tools/pattern_probe.hip) takes 0.31 ms per 32 x 4K: "
                    "arithmetic and memory are near-equal and overlap imperfectly" %
                    (100 * e["valu_busy_frac"], e["valu_insts_per_wave"], e["cycles_per_valu_inst"], 100 * e.get("tcp_pending_stall_frac", 0), 100 * e["ta_busy_frac"],
                     100 * e["lds_busy_frac"], 100 * e["lds_bank_conflict_share"], e["hbm_bytes_per_launch"] / alg))
    doc[key] = e
others = {}
for k in sorted(summ):
    name, grid = k.rsplit("|", 1)
    if "rwh::warp" not in name and "rwh::stitch" not in name:
        continue
    if name == KERNEL and int(grid) in (31211520, 28483584):
        continue
    try:
        f = facts(name, int(grid))
    except (KeyError, ZeroDivisionError, ValueError):
        continue
    # TCC_BUSY counts cycles with ANY request pending (~100 % for every streaming kernel): not a throughput figure.  What tells a
    # memory-bound kernel apart is how long the CUs' L1s sit on outstanding L2 / HBM requests, and the HBM rate it reaches
    # (a plain copy of read + write bytes gets 4.7-4.9 TB/s from this memory system: profiles/r04_lab_notes.txt section 5)
    busy = {"VALU": f["valu_busy_frac"], "texture-address path": f["ta_busy_frac"], "LDS": f["lds_busy_frac"]}
    top = sorted(busy.items(), key=lambda kv: -kv[1])
    f["hbm_TBps_in_pmc_pass"] = round(f["hbm_bytes_per_launch"] / f["kernel_us_in_pmc_pass"] / 1e6, 2)
    f["l2_requests_per_channel_clock"] = round(f.get("tcc_requests", 0) / 128.0 / f["kernel_cycles"], 3)
    f["limiter"] = ("%s: HBM traffic %.2f TB/s (a plain copy: 4.7-4.9), L1s waiting on outstanding L2 / HBM requests %d %% of the cycles, "
                    "L2 request rate %.2f per channel-clock; execution units: %s" %
                    ("memory system" if f.get("tcp_pending_stall_frac", 0) > 0.6 and top[0][1] < 0.95 else "%s and the memory system" % top[0][0],
                     f["hbm_TBps_in_pmc_pass"], 100 * f.get("tcp_pending_stall_frac", 0), f["l2_requests_per_channel_clock"],
                     ", ".join("%s %d %%" % (n, 100 * v) for n, v in top)))
    others["%s | grid %s" % (name.replace("void ", ""), grid)] = f
doc["other_kernels"] = others
json.dump(doc, open(out, "w"), indent=0)
print("wrote", out, "entries:", [k for k in doc if k != "other_kernels"], "+", len(others), "other kernels")
