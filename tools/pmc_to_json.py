"""Turn the summary of tools/pmc_pass.sh (summary.json + the kernel traces of its passes) into profiles/r03_pmc.json's two
entries for the headline kernel (4K x 32 and 8K x 8 launches).  usage: python tools/pmc_to_json.py <pmc dir> <out json>"""
import csv, glob, json, os, statistics, sys
root, out = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(root, "summary.json")))
KERNEL = "void rwh::warp_rgb8_fast8<unsigned char, 6>(rwh::FastArgs)"
def kernel_us(grid):
    f = glob.glob(os.path.join(root, "pass1", "**", "*kernel_trace.csv"), recursive=True)[0]
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
         if r["Kernel_Name"] == KERNEL and r.get("Grid_Size", r.get("Grid_Size_X")) == str(grid)]
    return statistics.mean(v[len(v) // 2:])
doc = json.load(open(out)) if os.path.exists(out) else {}
for key, grid, src, frames, alg in (("rwh::warp_rgb8_fast8<unsigned char, 6>", 31211520, [3840, 2160], 32, 1530430848),
                                    ("rwh::warp_rgb8_fast8<unsigned char, 6> @ 8 x 7680x4320", 28483584, [7680, 4320], 8, 1476410496)):
    c = summ["%s|%d" % (KERNEL, grid)]
    cyc = c["SQ_BUSY_CYCLES"] / 32                      # 32 shader engines
    us = kernel_us(grid)
    e = {"src": src, "frames": frames, "waves_per_launch": round(c["SQ_WAVES"]),
         "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1), "salu_insts_per_wave": round(c["SQ_INSTS_SALU"] / c["SQ_WAVES"], 1),
         "kernel_cycles": round(cyc), "sclk_hz": round(cyc / (us * 1e-6)), "kernel_us_in_pmc_pass": round(us, 1),
         "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 3), "cycles_per_valu_inst": round(c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"], 3),
         "ta_busy_frac": round(c["TA_BUSY_avr"] / cyc, 3), "lds_busy_frac": round(c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 3),
         "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3),
         "vmem_rd_insts_per_wave": round(c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], 2), "vmem_wr_insts_per_wave": round(c["SQ_INSTS_VMEM_WR"] / c["SQ_WAVES"], 2),
         "FETCH_SIZE_KB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KB_raw": c["WRITE_SIZE"],
         "hbm_read_bytes_per_launch": round(c["FETCH_SIZE"] * 2048), "hbm_write_bytes_per_launch": round(c["WRITE_SIZE"] * 1024),
         "algorithmic_bytes_per_launch": alg}
    e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    e["limiter"] = ("co-limited: VALU busy %d %% (%d VALU instructions per 512-pixel wave at %.2f cycles each), texture-address path busy %d %%, "
                    "LDS busy %d %% (%d %% of it bank conflicts); HBM traffic = %.4f x algorithmic" %
                    (round(e["valu_busy_frac"] * 100), round(e["valu_insts_per_wave"]), e["cycles_per_valu_inst"], round(e["ta_busy_frac"] * 100),
                     round(e["lds_busy_frac"] * 100), round(e["lds_bank_conflict_share"] * 100), e["hbm_bytes_per_launch"] / alg))
    doc[key] = e
    print(key, e["limiter"])
json.dump(doc, open(out, "w"), indent=0)
