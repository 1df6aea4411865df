"""Loop finder shared by the ISA report tools."""

def find_main_loops(body, lo=3000, hi=8000):
    """the CG loops of a kernel body: backward branches spanning lo..hi lines, one per disjoint region (latest start, furthest end)"""
    import re
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = set()
    for i, l in enumerate(body):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i and lo <= i - labels[m.group(2)] <= hi:
            a0 = labels[m.group(2)]
            if not any(t.strip().startswith("s_endpgm") for t in body[a0:i]):      # a span with an exit in it is not a loop body
                loops.add((a0, i))
    groups = []
    for a, b in sorted(loops):
        for g in groups:
            if not (b < g[0][0] or a > max(e for _, e in g)):
                g.append((a, b)); break
        else:
            groups.append([(a, b)])
    out = []
    for g in groups:
        # loops that share the CG loop's back edge region: take the common innermost start among the long ones
        a = max(s for s, e in g if e - s >= 0.9 * max(e2 - s2 for s2, e2 in g))
        b = max(e for s, e in g if s == a)
        out.append((a, b))
    return out


HALF = ("v_pk_", "v_bfe_", "_dpp", "f64", "v_readlane", "v_writelane", "v_readfirstlane", "v_rcp", "v_mul_lo", "v_mul_hi", "v_cvt_f64", "v_mad_u64", "v_div_")


def issue_clocks(t):
    """SIMD clocks one wave64 instruction occupies with two waves per SIMD (profiles/r02_valu_issue.json, measured by
    tools/valu_issue_bench.hip): 2.3 at full rate; 4.5 for packed fp32, v_bfe, DPP, fp64, lane reads / writes -- and for any
    instruction that reads a scalar register (v_cndmask with its mask in an SGPR pair, v_fmac with a scalar factor, ...)"""
    import re
    ins = t.split()[0]
    if not ins.startswith("v_"):
        return 0.0
    ops = t[len(ins):]
    src = ops.split(",", 1)[1] if "," in ops else ""
    if any(k in t for k in HALF) or re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc\b", src):
        return 4.5
    return 2.3
