#!/usr/bin/env python
"""Per-trip instruction counts of the two compositing loops, read from the compiler's gfx950 ISA.

usage: python scripts/isa_loop_counts.py [--show]        (runs here: hipcc cross-compiles without a GPU)

Compiles casualhdrsplat_amd/csrc/render.hip to device assembly with the flags of the Makefile, finds in
render_fwd_kernel<false,false> and render_bwd_kernel<false,false> the innermost loop that contains v_exp_f32 (the
per-(wave, entry) trip of the front-to-back / back-to-front replay), and counts its instructions by kind.  Where the
loop has an early `continue` (the backward skips the replay of an entry no lane is active for) the count is given for
both paths.  A loop body that holds several trips (the forward's holds two: two pairs of v_exp_f32) is divided by
that number; `full_trip` keeps the counts of the whole body.  Issue cycles weight each vector instruction by its measured wave64 issue cost on MI355X
(profiles/r01b_valu_rate.txt): 4 cycles, 8 for transcendentals and v_permlane*_swap.

Writes profiles/isa_loop_counts.json with the SHA-256 of the source it was made from; bench.py multiplies these
counts by the trip counts it measures live (hs_render_stats) to place the kernels on the VALU-issue roofline, and
reports `isa_stale` when render.hip has changed since.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "casualhdrsplat_amd", "csrc", "render.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def kernel_source_hash(path):
    """Same rule as bench.kernel_source_hash: blank and comment-only lines do not count."""
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for ln in f:
            t = ln.strip()
            if t and not t.startswith(b"//"):
                h.update(t + b"\n")
    return h.hexdigest()


def device_asm() -> str:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "render.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                               "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def functions(asm: str) -> dict:
    fns, name, body = {}, None, []
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                fns[name] = body
                name = None
            else:
                body.append(ln)
    return fns


def parse(body):
    """-> list of ('label', name) / ('ins', mnemonic, operands-text)"""
    items = []
    for ln in body:
        code = ln.split(";")[0].rstrip()
        if not code.strip():
            continue
        m = re.match(r"^(\.LBB\w+):", code)
        if m:
            items.append(("label", m.group(1)))
            continue
        t = code.strip()
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        items.append(("ins", parts[0], parts[1] if len(parts) > 1 else ""))
    return items


def loops_of(items):
    labels = {it[1]: i for i, it in enumerate(items) if it[0] == "label"}
    out = []
    for i, it in enumerate(items):
        if it[0] == "ins" and it[1].startswith(("s_cbranch", "s_branch")):
            tgt = it[2].strip()
            if tgt in labels and labels[tgt] < i:
                out.append((labels[tgt], i))
    return out


def has(items, lo, hi, what):
    return any(x[0] == "ins" and what in x[1] for x in items[lo:hi + 1])


def hot_loops(items, marker):
    """(full-trip loop, empty-trip loop or None).  The trip loop is the innermost loop containing `marker` (an
    instruction only the full trip executes), extended to the last back-branch to its head (the compiler emits one
    per `continue`); when a smaller loop with v_exp_f32 sits inside it, that one is the path of a trip that finds no
    active lane (head .. test .. back to the head)."""
    loops = loops_of(items)
    with_marker = [l for l in loops if has(items, l[0], l[1], marker)]
    if not with_marker:
        raise SystemExit(f"no loop containing {marker}")
    head = min(with_marker, key=lambda l: l[1] - l[0])[0]
    full = (head, max(l[1] for l in loops if l[0] == head))
    inner = [l for l in loops if full[0] <= l[0] and l[1] <= full[1] and l != full and l[0] != head
             and has(items, l[0], l[1], "v_exp_f32")]
    empty = min(inner, key=lambda l: l[1] - l[0]) if inner else None
    return full, empty


def count(seq):
    c = {"valu": 0, "valu_trans": 0, "valu_permlane_swap": 0, "valu_dpp": 0, "valu_packed": 0, "salu": 0, "lds": 0,
         "vmem": 0, "s_nop_wait_states": 0, "s_waitcnt": 0, "branch": 0}
    for it in seq:
        if it[0] != "ins":
            continue
        m, ops = it[1], it[2]
        if m.startswith("v_"):
            c["valu"] += 1
            if m.startswith(TRANS):
                c["valu_trans"] += 1
            if "permlane" in m and "swap" in m:
                c["valu_permlane_swap"] += 1
            if "dpp" in m or "row_" in ops or "quad_perm" in ops:
                c["valu_dpp"] += 1
            if m.startswith("v_pk_"):
                c["valu_packed"] += 1
        elif m == "s_nop":
            c["s_nop_wait_states"] += int(ops.strip() or 0) + 1
        elif m.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif m.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif m.startswith("s_"):
            c["salu"] += 1
        elif m.startswith("ds_"):
            c["lds"] += 1
        elif m.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    c["valu_issue_cycles"] = 4 * c["valu"] + 4 * (c["valu_trans"] + c["valu_permlane_swap"])
    return c


def analyse(items, marker):
    full_rng, empty_rng = hot_loops(items, marker)
    loop = items[full_rng[0]:full_rng[1] + 1]
    full = count(loop)
    # a trip evaluates two exponentials (one per pixel of the lane); the forward's loop body holds two trips
    tpi = max(1, sum(1 for x in loop if x[0] == "ins" and x[1].startswith("v_exp_f32")) // 2)
    res = {"trips_per_iteration": tpi, "valu_per_trip": full["valu"] / tpi,
           "valu_cycles_per_trip": full["valu_issue_cycles"] / tpi, "salu_per_trip": full["salu"] / tpi,
           "full_trip": full, "loop_instructions": sum(1 for x in loop if x[0] == "ins")}
    if empty_rng is not None:
        short = count(items[empty_rng[0]:empty_rng[1] + 1])
        res.update({"valu_per_empty_trip": short["valu"], "valu_cycles_per_empty_trip": short["valu_issue_cycles"],
                    "empty_trip": short})
    return res, loop


def _vregs(tok):
    """Register numbers named by one operand token: v12 -> {12}, v[4:7] -> {4,5,6,7}, anything else -> {}."""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_dpp_hazards(fns):
    """The hand-placed wait states of render.hip's inline-asm DPP blocks (reduce_in_rows / reduce_in_quads), checked in
    the ISA the compiler actually emitted around them: gfx9 needs TWO wait states between a VALU write of a VGPR and a
    DPP read of it as the permuted source (src0), and the hazard recogniser does not look inside inline asm.  Every DPP
    instruction of every function is checked against the instructions laid out before it (one wait state each, `s_nop k`
    = k + 1; layout order, which is what a fall-through sees -- the asm blocks open with their own s_nop for every
    other way in).  Returns (number of DPP instructions, violations)."""
    n_dpp, bad = 0, []
    for name, body in fns.items():
        items = [it for it in parse(body) if it[0] == "ins"]
        for i, it in enumerate(items):
            m, ops = it[1], it[2]
            if not (m.startswith("v_") and ("_dpp" in m or "row_" in ops or "quad_perm" in ops or "wave_" in ops)):
                continue
            n_dpp += 1
            toks = ops.split(",")
            src0 = _vregs(toks[1].split()[0]) if len(toks) > 1 else set()
            waited, j = 0, i - 1
            while j >= 0 and waited < 2:
                pm, pops = items[j][1], items[j][2]
                if pm == "s_nop":
                    waited += int(pops.strip() or 0) + 1
                else:
                    if pm.startswith("v_") and not pm.startswith(("v_cmp", "v_cmpx")):
                        if _vregs(pops.split(",")[0]) & src0:
                            bad.append(f"{name}: `{m} {ops}` reads v{sorted(src0)} {waited} wait state(s) after `{pm} {pops}`")
                    waited += 1
                j -= 1
    return n_dpp, bad


def main():
    asm = device_asm()
    fns = functions(asm)
    n_dpp, bad = check_dpp_hazards(fns)
    if bad:
        raise SystemExit("DPP read-after-VALU-write hazard (2 wait states needed):\n  " + "\n  ".join(bad[:20]))
    out = {"source": "scripts/isa_loop_counts.py (hipcc -O3 -ffp-contract=fast -S --cuda-device-only, gfx950)",
           "render_hip_sha256": kernel_source_hash(SRC),
           "issue_cycle_model": "4 cycles per wave64 vector instruction, 8 for transcendental and v_permlane*_swap "
                                "(profiles/r01b_valu_rate.txt)",
           "dpp_hazard_check": {"dpp_instructions_checked": n_dpp, "violations": 0,
                                "rule": "VALU write of a VGPR -> DPP read of it as src0: >= 2 wait states, over every "
                                        "function of the emitted ISA (the inline-asm blocks carry their own s_nop)"}}
    for kern in ("render_fwd_kernel", "render_bwd_kernel"):
        cands = [n for n in fns if kern in n and "ILb0ELb0E" in n]
        if len(cands) != 1:
            raise SystemExit(f"{kern}: expected one <false,false> instantiation, found {cands}")
        res, loop = analyse(parse(fns[cands[0]]), "v_exp_f32" if kern == "render_fwd_kernel" else "v_rcp_f32")
        res["symbol"] = cands[0]
        out[kern] = res
        if "--show" in sys.argv:
            print(f"==== {kern}: hot loop ({res['loop_instructions']} instructions)")
            for x in loop:
                print("   ", x[1] + ":" if x[0] == "label" else f"    {x[1]} {x[2]}")
    path = os.path.join(ROOT, "profiles", "isa_loop_counts.json")
    json.dump(out, open(path, "w"), indent=1)
    for kern in ("render_fwd_kernel", "render_bwd_kernel"):
        r = out[kern]
        print(kern, "VALU/trip", r["valu_per_trip"], "cycles/trip", r["valu_cycles_per_trip"], "empty-trip VALU",
              r.get("valu_per_empty_trip"), {k: v for k, v in r["full_trip"].items() if k != "valu"})


if __name__ == "__main__":
    main()
