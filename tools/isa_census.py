#!/usr/bin/env python3
"""ISA census of a fused kernel's frame loop, from the compiler's own assembly (no GPU needed).

  python tools/isa_census.py group    [-o profiles/r4_isa_synth_kernel_group.txt]
  python tools/isa_census.py dual     [-o profiles/r4_isa_synth_dual_kernel.txt]
  python tools/isa_census.py --unit synth_dual.hip --kernel 'synth_dual_kernel<false, false, 0, false, false, 0>' ...

The translation unit is compiled for the device only with `-S -gline-tables-only` (line tables do not change the code:
the instruction count is the same with and without), so every instruction carries its source position WITH its inlining
chain.  The frame loop is the kernel's `for (it < iters)` loop: every instruction whose call site lies in its source range.  An instruction belongs to the PHASE
whose VPZ_STAMP interval holds the outermost position of its chain that lies in the kernel's own body (the call site in
the loop), and to the helper FUNCTION of its innermost position.

What is counted is static: instructions of the loop body, split into the all-long steady-state path ("hot": what a
2048-after-2048 frame with long windows executes) and the rest ("cold": batches of short blocks, the 256-point
transform, unaligned / trimmed emission, the integer DDA, tuning switches), by source region -- see COLD below.
Every s_waitcnt of the hot path is listed with its counters and the memory instructions issued since the last wait of
the same counter (so an lgkmcnt wait that covers a scalar load shows as such)."""
import argparse
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "vorbispizza_amd", "csrc")

PRESETS = {
    "group": ("synth_kernels.hip", "synth_kernel<true, 0, false, true, false>"),
    "group_ilv": ("synth_kernels.hip", "synth_kernel<true, 1, false, true, false>"),
    "dual": ("synth_dual.hip", "synth_dual_kernel<true, true, 1, false, false, 0>"),
    "dual_nofloor": ("synth_dual.hip", "synth_dual_kernel<false, false, 0, false, false, 0>"),
    "pairs": ("synth_pairs.hip", "synth_pairs_kernel<true, true, 0, false, false, 0>"),
    "pairs_ilv": ("synth_pairs.hip", "synth_pairs_kernel<true, true, 1, false, false, 0>"),
}
# a unit that is another one compiled once more under a macro: (the file its kernel's text is in, the kernel's name there)
SOURCE_OF = {"synth_pairs.hip": ("synth_dual.hip", "synth_dual_kernel")}

# Source regions that an all-long floored frame does not execute.  Functions by name (innermost or anywhere in the
# inlining chain); kernel-body regions by a `// [census: cold]` mark on the line that opens their brace block.
COLD_FUNCTIONS = [
    r"imdct256_wave8", r"imdct_mid_wave", r"render_floor_indices$", r"render_floor_indices<", r"unpack_segment",
    r"div_floor_small", r"pickup_interleaved", r"stage_by_lds_dma", r"y_from_h", r"tail_at", r"clip_value", r"clip_track",
    r"to_s16", r"pack_s16",
]
COLD_MARK = "[census: cold]"  # on the line that opens a brace block the all-long steady state does not enter


def compile_asm(unit, out_dir):
    from vorbispizza_amd import _build
    src = os.path.join(CSRC, unit)
    dst = os.path.join(out_dir, unit.replace(".hip", ".s"))
    stale = not os.path.exists(dst) or any(
        os.path.getmtime(os.path.join(CSRC, f)) > os.path.getmtime(dst) for f in os.listdir(CSRC))
    if stale:
        extra = os.environ.get("VPZ_EXTRA_HIPCC_FLAGS", "").split()
        cmd = [_build._hipcc()] + _build.COMMON + _build.SOURCES[unit] + extra + [
            "--cuda-device-only", "-S", "-gline-tables-only", "-o", dst, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return dst


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True, check=True).stdout
    return dict(zip(names, out.split("\n")))


def source_functions(path):
    """[(first_line, last_line, name)] of the free functions and `auto name = [&](...)` lambdas of a source file."""
    lines = open(path).read().split("\n")
    funcs = []
    i = 0
    while i < len(lines):
        l = lines[i]
        m = None
        if re.match(r"^(static |inline |template|__device__|__global__|__host__|hipError_t|bool |int |void |size_t )", l) and "(" in l and not l.rstrip().endswith(";"):
            j = i
            head = l
            while "{" not in lines[j] and j + 1 < len(lines) and not lines[j].rstrip().endswith(";"):
                j += 1
                head += " " + lines[j]
            m = re.search(r"([A-Za-z_]\w*)\s*\(", re.sub(r"__launch_bounds__\([^)]*\)(\s*\))?", "", re.sub(r"template\s*<[^>]*>", "", head)))
            if m and "{" in lines[j]:
                k = j
                while k < len(lines) and not lines[k].startswith("}"):
                    k += 1
                funcs.append((i + 1, k + 1, m.group(1)))
        i += 1
    # lambdas (indented, `auto name = [&]`)
    for i, l in enumerate(lines):
        m = re.match(r"^(\s+)auto (\w+) = \[[&=]?\]", l)
        if m:
            ind = m.group(1)
            k = i
            if l.rstrip().endswith("};"):  # a one-line lambda
                funcs.append((i + 1, i + 1, m.group(2)))
                continue
            while k < len(lines) and not lines[k].startswith(ind + "};"):
                k += 1
            funcs.append((i + 1, k + 1, m.group(2)))
    return funcs


def brace_block_end(lines, start):
    """Index of the line that closes the brace block opened on lines[start] (the last `{` of that line)."""
    depth = 0
    opened = False
    # (preprocessor alternatives: `#ifdef X / if (a) { / #else / if (b) { / #endif` opens ONE block -- only the branch a build
    # without -D switches compiles is counted: the #else branch of an #ifdef, the first branch of an #ifndef)
    active = [True]
    for k in range(start, len(lines)):
        pp = lines[k].strip()
        if pp.startswith("#ifdef") or (pp.startswith("#if ") and "defined" in pp and "!defined" not in pp):
            active.append(False)
            continue
        if pp.startswith("#ifndef") or pp.startswith("#if"):
            active.append(True)
            continue
        if pp.startswith("#else") and len(active) > 1:
            active[-1] = not active[-1]
            continue
        if pp.startswith("#elif") and len(active) > 1:
            active[-1] = False
            continue
        if pp.startswith("#endif") and len(active) > 1:
            active.pop()
            continue
        if not all(active):
            continue
        code = re.sub(r'//.*', "", lines[k])
        if k == start:  # the block is the one the LAST `{` of the opening line opens (`} else if (...) {`)
            code = code[code.rfind("{"):]
        for ch in code:
            if ch == "{":
                depth += 1
                opened = True
            elif ch == "}":
                depth -= 1
                if opened and depth == 0:
                    return k
    return len(lines) - 1


CLASS_ORDER = ["VALU", "VALU_xlane", "SALU", "SMEM", "LDS", "VMEM_load", "VMEM_store", "FLAT", "branch", "waitcnt", "barrier", "other"]


def classify(op):
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_store", "s_dcache")):
        return "SMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_call")):
        return "branch"
    if op.startswith(("s_nop", "s_sleep", "s_setprio", "s_sethalt", "s_code_end", "s_trap")):
        return "other"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_load", "buffer_load", "scratch_load")):
        return "VMEM_load"
    if op.startswith(("global_store", "buffer_store", "scratch_store", "global_atomic", "buffer_atomic")):
        return "VMEM_store"
    if op.startswith("flat_"):
        return "FLAT"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")) or "dpp" in op:
        return "VALU_xlane"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("preset", nargs="?", choices=sorted(PRESETS))
    ap.add_argument("--unit")
    ap.add_argument("--kernel")
    ap.add_argument("-o", "--out")
    ap.add_argument("--tmp", default="/tmp/vpz_isa")
    ap.add_argument("--dump-loop", help="write the annotated hot path of the loop to this file")
    ap.add_argument("--csrc", help="another copy of vorbispizza_amd/csrc to analyse (an older commit's sources, with the cold marks added)")
    args = ap.parse_args()
    unit, kernel = PRESETS[args.preset] if args.preset else (args.unit, args.kernel)
    global CSRC
    if args.csrc:
        CSRC = os.path.abspath(args.csrc)
    os.makedirs(args.tmp, exist_ok=True)
    asm = compile_asm(unit, args.tmp)
    text = open(asm).read().split("\n")

    # ---- the kernel's body in the assembly
    labels = [(i, m.group(1)) for i, l in enumerate(text) for m in [re.match(r"^(_Z\w+):", l)] if m]
    names = demangle([n for _, n in labels])
    start = end = None
    for i, n in labels:
        d = names[n]
        if d.startswith("void vpz::" + kernel + "(") or d.startswith("vpz::" + kernel + "("):
            start = i
            break
    if start is None:
        sys.exit("kernel not found: " + kernel)
    for k in range(start, len(text)):
        if text[k].startswith(".Lfunc_end"):
            end = k
            break
    body = text[start:end]

    # ---- file table and the kernel's source
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = os.path.normpath(os.path.join(m.group(2), m.group(3)))
    src_unit = SOURCE_OF.get(unit, (unit, None))[0]
    unit_path = os.path.join(CSRC, src_unit)
    src_lines = open(unit_path).read().split("\n")
    kname = SOURCE_OF.get(unit, (None, kernel.split("<")[0]))[1]
    kfirst = next(i for i, l in enumerate(src_lines) if re.search(r"\bvoid " + kname + r"\(SynthArgs a\)", l)) + 1
    klast = next(i for i in range(kfirst, len(src_lines)) if src_lines[i].startswith("}")) + 1
    loop_line = next(i for i in range(kfirst, klast) if re.search(r"for \(int it = 0; it < iters; \+\+it\)", src_lines[i])) + 1
    loop_end_line = brace_block_end(src_lines, loop_line - 1) + 1
    stamps = []  # (line, k, text)
    for i in range(loop_line, loop_end_line):
        m = re.search(r"VPZ_STAMP\((\d+)\);\s*//\s*(.*)", src_lines[i])
        if m:
            stamps.append((i + 1, int(m.group(1)), m.group(2).strip()))

    def phase_of_line(line):
        if line < loop_line:
            return "(ahead of the loop: hoisted by the compiler)"
        if line > loop_end_line:
            return "(behind the loop)"
        for ln, k, t in stamps:
            if line <= ln:
                return "%d %s" % (k, t)
        return "%d loop bookkeeping" % (stamps[-1][1] + 1 if stamps else 0)

    func_tables = {}

    def func_of(path, line):
        if path not in func_tables:
            try:
                func_tables[path] = source_functions(path)
            except OSError:
                func_tables[path] = []
        best = None
        for a, b, n in func_tables[path]:
            if a <= line <= b and (best is None or a >= best[0]):
                best = (a, b, n)
        return best[2] if best else os.path.basename(path)

    cold_ranges = []
    for i in range(kfirst - 1, klast):
        if COLD_MARK in src_lines[i]:
            label = re.sub(r"\s*//.*", "", src_lines[i].strip())
            cold_ranges.append((i + 2, brace_block_end(src_lines, i) + 1, label))  # (the opening line's condition is evaluated)
    cold_fn = [re.compile(p) for p in COLD_FUNCTIONS]

    # ---- walk the kernel: labels, locations, instructions
    insts = []  # dict(idx, op, args, chain[(path, line)], label)
    cur_chain = []
    lab_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab_at[m.group(1)] = len(insts)
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)\s+\d+.*?;\s*(.*)$", l)
        if m:
            chain = []
            for pm in re.finditer(r"([^\s\[\]@]+):(\d+):\d+", m.group(3)):
                chain.append((os.path.normpath(pm.group(1)), int(pm.group(2))))
            if not chain:
                chain = [(files.get(int(m.group(1)), "?"), int(m.group(2)))]
            if int(m.group(2)) != 0:  # (line 0: code the compiler merged from several places -- stays with its neighbours)
                cur_chain = chain
            continue
        m = re.match(r"^\s+([a-z_][a-z0-9_]*)\s*(.*?)\s*(;.*)?$", l)
        if m and not l.strip().startswith("."):
            insts.append({"op": m.group(1), "args": m.group(2), "chain": cur_chain, "line": i})
    # the frame loop: largest back edge
    best = None
    for idx, ins in enumerate(insts):
        if ins["op"].startswith(("s_cbranch", "s_branch")):
            tgt = ins["args"].split()[-1] if ins["args"] else ""
            if tgt in lab_at and lab_at[tgt] <= idx and (best is None or idx - lab_at[tgt] > best[1] - best[0]):
                best = (lab_at[tgt], idx)
    if best is None:
        sys.exit("no loop found")
    lo, hi = best

    def attribute(ins):
        chain = ins["chain"]
        body_line = None
        for path, line in reversed(chain):  # outermost first
            if os.path.basename(path) == src_unit and kfirst <= line <= klast:
                body_line = line
                break
        inner_path, inner_line = chain[0] if chain else ("?", 0)
        fn = func_of(inner_path, inner_line) if os.path.exists(inner_path) else os.path.basename(inner_path)
        if fn == kname:
            fn = "(kernel body)"
        cold = None
        for path, line in chain:
            if os.path.basename(path) == src_unit:
                for a, b, anchor in cold_ranges:
                    if a <= line <= b:
                        cold = anchor
            if os.path.exists(path):
                f = func_of(path, line)
                if any(p.search(f) for p in cold_fn):
                    cold = f
        return phase_of_line(body_line) if body_line else "(no position in the kernel body)", fn, cold

    out = []
    w = out.append
    w("ISA census: %s :: %s" % (unit, kernel))
    w("assembly: hipcc %s -S -gline-tables-only (device only); kernel %d instructions, frame loop %d (instructions %d..%d)"
      % (" ".join(__import__("vorbispizza_amd._build", fromlist=["x"]).COMMON), len(insts), hi - lo + 1, lo, hi))
    res = {}
    for l in text[end:end + 400]:
        m = re.match(r"\s*[;.]\s*\.?(sgpr_count|vgpr_count|NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize|sgpr_spill_count|vgpr_spill_count|codeLenInByte)\s*[:=]?\s*(\S+)", l)
        if m and m.group(1) not in res:
            res[m.group(1)] = m.group(2)
    w("resources: " + ", ".join("%s %s" % kv for kv in res.items()))
    w("")

    per_phase = OrderedDict()
    per_fn = defaultdict(Counter)
    cold_tot = Counter()
    cold_by = defaultdict(Counter)
    hot_listing = []
    pend = {"vm": [], "lgkm": []}
    waits = []
    for idx in range(len(insts)):
        ins = insts[idx]
        phase, fn, cold = attribute(ins)
        # the loop's instructions by SOURCE position (the compiler lays some of the loop's blocks out behind its back edge);
        # code without a position in the kernel body belongs to the loop if it lies inside it
        if phase.startswith("(no position") and not lo <= idx <= hi:
            continue
        if phase.startswith("(ahead") or phase.startswith("(behind"):
            continue
        cls = classify(ins["op"])
        if cold:
            cold_tot[cls] += 1
            cold_by[cold][cls] += 1
            continue
        per_phase.setdefault(phase, Counter())[cls] += 1
        per_fn[(phase, fn)][cls] += 1
        hot_listing.append((idx, phase, fn, ins))
        if cls in ("VMEM_load", "VMEM_store"):
            pend["vm"].append(ins["op"])
        elif cls == "FLAT":
            pend["vm"].append(ins["op"])
            pend["lgkm"].append(ins["op"])
        elif cls in ("LDS", "SMEM"):
            pend["lgkm"].append(ins["op"])
        elif cls == "waitcnt":
            a = ins["args"]
            covered = []
            if "vmcnt" in a:
                covered.append("vm: " + (_summ(pend["vm"]) or "-"))
                if "vmcnt(0)" in a:
                    pend["vm"] = []
            if "lgkmcnt" in a:
                covered.append("lgkm: " + (_summ(pend["lgkm"]) or "-"))
                if "lgkmcnt(0)" in a:
                    pend["lgkm"] = []
            waits.append((phase, fn, a, "; ".join(covered)))

    def row(name, c):
        return "%-58s" % name[:58] + "".join("%8d" % c.get(k, 0) for k in CLASS_ORDER) + "%8d" % sum(c.values())

    hdr = "%-58s" % "" + "".join("%8s" % k[:7] for k in CLASS_ORDER) + "%8s" % "total"
    w("== frame loop, hot path (static instruction counts; an all-long floored frame with long windows), by phase ==")
    w(hdr)
    tot = Counter()
    for ph in sorted(per_phase, key=lambda s: (not s[0].isdigit(), int(s.split()[0]) if s[0].isdigit() else 0, s)):
        w(row(ph, per_phase[ph]))
        tot.update(per_phase[ph])
    w(row("TOTAL hot path", tot))
    salu, valu = tot["SALU"], tot["VALU"] + tot["VALU_xlane"]
    w("scalar ALU per vector ALU instruction: %.2f; scalar memory instructions per pass: %d" % (salu / max(1, valu), tot["SMEM"]))
    w("")
    w("== ... by phase and helper function ==")
    w(hdr)
    for (ph, fn) in sorted(per_fn, key=lambda t: (not t[0][0].isdigit(), int(t[0].split()[0]) if t[0][0].isdigit() else 0, t[0], -sum(per_fn[t].values()))):
        w(row("  %s | %s" % (ph.split(" ")[0], fn), per_fn[(ph, fn)]))
    w("")
    w("== cold code inside the loop (not executed by an all-long frame), by region ==")
    w(hdr)
    for k in sorted(cold_by, key=lambda k: -sum(cold_by[k].values())):
        w(row(k, cold_by[k]))
    w(row("TOTAL cold", cold_tot))
    w("")
    w("== every s_waitcnt of the hot path, in program order ==")
    w("%-44s %-26s %-28s %s" % ("phase", "function", "counters", "memory instructions issued since the counter was last waited to 0"))
    for ph, fn, a, cov in waits:
        w("%-44s %-26s %-28s %s" % (ph[:44], fn[:26], a, cov))
    n_smem_waits = sum(1 for _, _, a, cov in waits if "lgkm" in a and "s_load" in cov)
    w("")
    w("waits: %d in the hot path; %d of them are lgkmcnt waits with a scalar load outstanding (they drain the LDS queue too)"
      % (len(waits), n_smem_waits))
    w("")
    w("== scalar loads of the hot path ==")
    for idx, ph, fn, ins in hot_listing:
        if classify(ins["op"]) == "SMEM":
            w("%-44s %-26s %s %s" % (ph[:44], fn[:26], ins["op"], ins["args"]))
    report = "\n".join(out) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(report)
        print("wrote", args.out)
    else:
        sys.stdout.write(report)
    if args.dump_loop:
        with open(args.dump_loop, "w") as f:
            for idx, ph, fn, ins in hot_listing:
                f.write("%6d  %-28s %-24s %s %s\n" % (idx, ph[:28], fn[:24], ins["op"], ins["args"]))


def _summ(ops):
    c = Counter()
    for o in ops:
        c[o.split("_")[0] + "_" + (o.split("_")[1] if "_" in o else "")] += 1
    return ", ".join("%d %s" % (v, k) for k, v in c.items())


if __name__ == "__main__":
    main()
