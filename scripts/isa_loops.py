"""Instruction mix of a kernel's loops from its gfx950 disassembly (runs without a GPU):
    scripts/isa_loops.py <csrc file> <substring of the mangled kernel name> [min loop length]
Lists every backward branch (loop) with its body's instruction counts by class: VALU (full-rate / half-rate / transcendental by
scripts/micro/valu_rate.hip's table), packed, DPP, LDS reads / writes / atomics, SALU, VMEM, waits."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
                       "--cuda-device-only", "-c", os.path.join(root, "map-merge_amd/csrc", src), "-o", tmp + "/d.o"] + sys.argv[4:])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + tmp + "/d.o",
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/d.co"])
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", tmp + "/d.co"], capture_output=True, text=True).stdout
HALF = ("v_max_", "v_min_", "v_mbcnt", "v_cvt", "v_mul_lo", "v_mul_hi", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_perm", "v_mad_u", "v_mad_i", "v_alignb", "v_ldexp",
        "v_frexp", "v_floor", "v_trunc", "v_rndne", "v_ceil", "v_fract", "v_med3", "v_min3", "v_max3", "v_readlane", "v_readfirstlane", "v_writelane", "v_add_co", "v_addc", "v_sub_co",
        "v_subb", "v_lshlrev_b64", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cmp_class", "v_div_", "v_mov_b32_dpp")
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
funcs = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
for f in funcs:
    head = f.split("\n", 1)[0]
    if pat not in head:
        continue
    print(head[:200])
    ins = []
    for line in f.split("\n")[1:]:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):(.*)", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    def classify(op, args):
        if op.startswith("v_pk_"): return "pk"
        if "dpp" in op or "quad_perm" in args or "row_" in args: return "dpp"
        if op.startswith("ds_"):
            if "read" in op or "load" in op: return "lds_rd"
            if "write" in op or "store" in op: return "lds_wr"
            return "lds_atom/other"
        if op.startswith(TRANS): return "trans"
        if op.startswith(HALF): return "half"
        if op.startswith("v_"): return "valu"
        if op.startswith("s_waitcnt"): return "wait"
        if op.startswith("s_"): return "salu"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
        return "other"
    total = {}
    for _, op, args in ins:
        c = classify(op, args); total[c] = total.get(c, 0) + 1
    print("  whole kernel:", len(ins), "instructions", dict(sorted(total.items())))
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<.*\+0x([0-9a-f]+)>", args)
            if not m: continue
            # target = function start + offset
            tgt = int(re.match(r"([0-9a-f]+)", head).group(1), 16) + int(m.group(1), 16)
            if tgt in addr_index and addr_index[tgt] < i and i - addr_index[tgt] >= minlen:
                body = ins[addr_index[tgt]:i + 1]
                cnt = {}
                for _, o, ar in body:
                    c = classify(o, ar); cnt[c] = cnt.get(c, 0) + 1
                print(f"  loop {addr_index[tgt]:5d} .. {i:5d} ({len(body):4d} instr):", dict(sorted(cnt.items())))
