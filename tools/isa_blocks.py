"""Developer tool: basic blocks of one kernel in hipcc's -S output with instruction-class counts.
usage: python tools/isa_blocks.py file.s SUBSTRING_OF_KERNEL_NAME [min_instrs]"""
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_sub_f32", "v_subrev_f32", "v_add_f32", "v_or_b32", "v_and_b32",
        "v_xor_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_add_co_u32",
        "v_addc_co_u32", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32"}
TRANS = {"v_log_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32",
         "v_rcp_iflag_f32"}


def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith(key) and ":" in l[:len(key) + 400].split(";")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []
    cur = [lines[start][:60], []]
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append(cur)
            cur = [s.split(":")[0], []]
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur[1].append(s.split(";")[0].strip())
        # a branch ends the basic block even where no label follows (the fall-through part gets the name + "'")
        if cur[1][-1].split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm")):
            blocks.append(cur)
            cur = [cur[0].rstrip("'") + "'", []]
    blocks.append(cur)
    tot = {}
    for name, ins in blocks:
        c = {"valu_fast": 0, "valu_slow": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "branch": 0}
        targets = []
        for i in ins:
            op = i.split()[0]
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
            if op.startswith("v_"):
                if base in TRANS:
                    c["trans"] += 1
                elif base in FAST and not op.endswith(("_sdwa", "_dpp")):
                    c["valu_fast"] += 1
                else:
                    c["valu_slow"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith(("s_cbranch", "s_branch")):
                c["branch"] += 1
                targets.append(i.split()[-1])
            elif op.startswith("s_"):
                c["salu"] += 1
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        n = sum(c.values())
        if n >= minn:
            cost = c["valu_fast"] * 1.15 + c["valu_slow"] * 1.85 + c["trans"] * 3.45
            print("%-12s n=%4d fast=%3d slow=%3d trans=%2d salu=%3d lds=%3d vmem=%3d wait=%2d  valu_ns=%6.1f -> %s" %
                  (name, n, c["valu_fast"], c["valu_slow"], c["trans"], c["salu"], c["lds"], c["vmem"], c["wait"],
                   cost, " ".join(targets)))
    print("total", tot)


main()
