"""Instruction histogram of one kernel's STAGED hot path from `hipcc -S` device assembly (developer tool).

   hipcc -O3 ... --cuda-device-only -S -o warp.s ransac_with_homography_amd/csrc/rwh_warp.hip
   python tools/isa_histogram.py warp.s _ZN3rwh15warp_rgb8_fast8IhLi6EEEvNS_8FastArgsE

The hot path = straight-line code from the kernel entry to the second full-width store (global_store_dwordx3), minus the
basic blocks of the ragged-row stores (the ones holding global_store_byte / global_store_short), which full tiles skip.
Phases are split at the landmarks the kernel's source has: last staging load, last staging LDS write, first store."""
import collections, re, sys
path, sym = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3 and sys.argv[3] == "--blocks":
    # per basic block: VALU / SALU / LDS / VMEM / scratch counts (since the border path joined the kernel, the straight-line
    # walk below no longer isolates the staged path: its two runs are the blocks with 8 LDS reads, 48 byte converts and no
    # 64-bit compares)
    text = open(path).read()
    a = text.index(sym + ":"); b = text.index(".Lfunc_end", a)
    name, cur, blocks = "entry", [], []
    for l in text[a:b].split("\n"):
        l = l.strip()
        if not l or l.startswith(";"): continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur)); name, cur = l, []
        else:
            cur.append(l)
    blocks.append((name, cur))
    for name, bl in blocks:
        c = collections.Counter(m.group(0) for m in (re.match(r"^(v_|s_|ds_|global_|scratch_)\S+", l) for l in bl) if m)
        n = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))
        if n("v_") > 10 or n("scratch_"):
            print("%-12s VALU %3d SALU %3d LDS %2d VMEM %2d scratch %d  byte converts %2d  u64 compares %d  writelane %d" %
                  (name, n("v_"), n("s_"), n("ds_"), n("global_"), n("scratch_"), sum(v for k, v in c.items() if "cvt_f32_ubyte" in k),
                   sum(v for k, v in c.items() if "_u64" in k and k.startswith("v_cmp")), c.get("v_writelane_b32", 0)))
    sys.exit(0)
if len(sys.argv) > 4 and sys.argv[3] == "--path":
    # histogram of an explicit list of basic blocks (round 3: pick the staged path's blocks from the --blocks listing:
    # the entry blocks, the staging-load and LDS-write blocks, the two run blocks with 8 LDS reads and no 64-bit compares)
    # "9:81" = block 9 up to its branch to block 81 (a labelled block may hold several branches; the code behind the taken one is not on the path)
    want = [w.split(":") for w in sys.argv[4].split(",")]
    text = open(path).read()
    a = text.index(sym + ":"); b = text.index(".Lfunc_end", a)
    name, cur, blocks = "entry", [], {}
    for l in text[a:b].split("\n"):
        l = l.strip()
        if not l or l.startswith(";"): continue
        m = re.match(r"^\.LBB\d+_(\d+):", l)
        if m: blocks[name] = cur; name, cur = m.group(1), []
        else: cur.append(l)
    blocks[name] = cur
    tot = collections.Counter()
    for w in want:
        bl = blocks[w[0]]
        if len(w) > 1:
            cut = next(i for i, l in enumerate(bl) if re.match(r"^s_c?branch\S* \.LBB\d+_%s$" % w[1], l))
            bl = bl[:cut + 1]
        tot.update(m.group(0) for m in (re.match(r"^(v_|s_|ds_|global_)\S+", l) for l in bl) if m)
    valu = sum(v for k, v in tot.items() if k.startswith("v_"))
    print("blocks %s: VALU %d per wave = %.2f per pixel (8 px per lane); SALU %d; LDS %d; VMEM %d" %
          (sys.argv[4], valu, valu / 8, sum(v for k, v in tot.items() if k.startswith("s_")),
           sum(v for k, v in tot.items() if k.startswith("ds_")), sum(v for k, v in tot.items() if k.startswith("global_"))))
    groups = [("blend (v_fma_mix_f32 on float16 tap halves)", r"v_fma_mix_f32"), ("tap halves (v_perm_b32) + RGB -> RGBX expansion", r"v_perm_b32|v_alignbyte"),
              ("weights (v_cvt_f32_u32, v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32, v_mul_f32, v_fma_f32)", r"v_cvt_f32_u32|v_pk_mul_f32|v_pk_fma_f32|v_sub_f32|v_mul_f32|v_fma_f32|v_fmac_f32|v_pk_add_f32"),
              ("convert + pack (v_cvt_pk_u8_f32, v_or3 / v_lshlrev / v_and of the 12-byte store words)", r"v_cvt_pk_u8_f32|v_or3_b32|v_and_or_b32"),
              ("float64 coordinates (v_*_f64)", r"_f64"),
              ("tap / staging addresses (24-bit multiplies, shifts, adds, min)", r"v_mul_u32_u24|v_mad_u32_u24|v_lshl_add_u32|v_lshlrev_b32|v_lshrrev_b32|v_add_u32|v_sub_u32|v_min_u32|v_min_i32|v_mul_lo_u32|v_mad_u64|v_lshl_add_u64|v_add3|v_mad_i32_i24|v_and_b32"),
              ("cross-lane (v_readlane / readfirstlane)", r"v_readlane|v_readfirstlane")]
    left = dict((k, v) for k, v in tot.items() if k.startswith("v_"))
    for gname, rx in groups:
        hit = [(k, v) for k, v in left.items() if re.search(rx, k)]
        n = sum(v for _, v in hit)
        for k, _ in hit: del left[k]
        print("   %-100s %3d  = %5.2f per pixel   (%s)" % (gname, n, n / 8, ", ".join("%s %d" % kv for kv in sorted(hit, key=lambda kv: -kv[1]))))
    n = sum(left.values())
    print("   %-100s %3d  = %5.2f per pixel   (%s)" % ("other", n, n / 8, ", ".join("%s %d" % kv for kv in sorted(left.items(), key=lambda kv: -kv[1]))))
    print("   memory: " + ", ".join("%s %d" % kv for kv in sorted(tot.items()) if kv[0].startswith(("ds_", "global_"))))
    sys.exit(0)
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
# basic blocks
blocks, cur = [], []
for l in body:
    if (re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb.")) and cur:
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
stores = [i for i, b in enumerate(blocks) if any("global_store_dwordx3" in l or "global_store_dwordx4" in l for l in b)]
last = stores[1] if len(stores) > 1 else stores[0]
hot = [b for b in blocks[:last + 1] if not any("global_store_byte" in l or "global_store_short" in l for l in b)]
flat = [l.strip() for b in hot for l in b]
def phase_of(idx, marks):
    for name, m in marks:
        if idx <= m: return name
    return marks[-1][0]
ld = max((i for i, l in enumerate(flat) if l.startswith("global_load_dwordx3")), default=0)
wr = max((i for i, l in enumerate(flat) if l.startswith("ds_write_b128")), default=ld)
st = [i for i, l in enumerate(flat) if l.startswith("global_store_dwordx")]
marks = [("1 prologue + end pixels + footprint + staging loads", ld), ("2 staging: RGB -> RGBX, LDS writes", wr),
         ("3 run 0: coordinates, weights, taps, blend, store", st[0] if st else len(flat)), ("4 run 1", len(flat))]
per = collections.OrderedDict((n, collections.Counter()) for n, _ in marks)
for i, l in enumerate(flat):
    m = re.match(r"^(v_|s_|ds_|global_|buffer_)\S+", l)
    if m: per[phase_of(i, marks)][m.group(0)] += 1
tot = collections.Counter()
for n, h in per.items():
    tot.update(h)
    valu = sum(v for k, v in h.items() if k.startswith("v_"))
    print("%-55s VALU %3d  SALU %3d  LDS %2d  VMEM %d" % (n, valu, sum(v for k, v in h.items() if k.startswith("s_")),
          sum(v for k, v in h.items() if k.startswith("ds_")), sum(v for k, v in h.items() if k.startswith("global_"))))
    print("      " + ", ".join("%s %d" % (k, v) for k, v in sorted(h.items(), key=lambda kv: -kv[1]) if k.startswith("v_")))
valu = sum(v for k, v in tot.items() if k.startswith("v_"))
print("TOTAL hot path: VALU %d per wave = %.2f per pixel (8 px per lane); SALU %d; LDS %d; VMEM %d" %
      (valu, valu / 8, sum(v for k, v in tot.items() if k.startswith("s_")), sum(v for k, v in tot.items() if k.startswith("ds_")),
       sum(v for k, v in tot.items() if k.startswith("global_"))))
groups = [("byte -> float converts (v_cvt_f32_ubyte*)", r"v_cvt_f32_ubyte"), ("blend (v_pk_fma_f32)", r"v_pk_fma_f32"),
          ("weights (v_cvt_f32_u32, v_pk_mul_f32, v_sub_f32, v_mul_f32, v_fma_f32, v_pk_add_f32)", r"v_cvt_f32_u32|v_pk_mul_f32|v_sub_f32|v_mul_f32|v_fma_f32|v_fmac_f32|v_pk_add_f32"),
          ("convert + pack (v_cvt_pk_u8_f32)", r"v_cvt_pk_u8_f32"), ("float64 coordinates (v_*_f64)", r"_f64"),
          ("tap / staging addresses (24-bit multiplies, shifts, adds, min)", r"v_mul_u32_u24|v_mad_u32_u24|v_lshl_add_u32|v_lshlrev_b32|v_lshrrev_b32|v_add_u32|v_sub_u32|v_min_u32|v_min_i32|v_mul_lo_u32|v_mad_u64|v_lshl_add_u64|v_add3"),
          ("RGB -> RGBX expansion (v_alignbyte_b32)", r"v_alignbyte"), ("cross-lane (v_readlane / readfirstlane)", r"v_readlane|v_readfirstlane")]
left = dict((k, v) for k, v in tot.items() if k.startswith("v_"))
for name, rx in groups:
    n = sum(v for k, v in list(left.items()) if re.search(rx, k))
    for k in [k for k in left if re.search(rx, k)]: del left[k]
    print("   %-95s %3d  = %5.2f per pixel" % (name, n, n / 8))
print("   %-95s %3d  = %5.2f per pixel   (%s)" % ("other", sum(left.values()), sum(left.values()) / 8, ", ".join("%s %d" % kv for kv in sorted(left.items(), key=lambda kv: -kv[1]))))
