"""Turns gpurun_out/prof_final/ (scripts/collect_profiles.sh) into the tracked files of profiles/ (development aid).
usage: python scripts/fold_profiles.py pmc|bench <round-tag> [c3|c2|c4]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
what, tag = sys.argv[1], sys.argv[2]
CFG = sys.argv[3] if len(sys.argv) > 3 else "c3"
SRC = os.path.join(ROOT, "gpurun_out", "prof_final" if CFG == "c3" else "prof_final_" + CFG)
DST = os.path.join(ROOT, "profiles")


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("hs::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def render_bwd_alone(sub, counter):
    """Average `counter` of the render_bwd_kernel launches WITHOUT the CRF gradient's tail workgroups in a
    `step_c3.py --render-bwd-alone` pass: the launches with the smallest grid (the first step of the pass is a whole
    backward, whose launch carries the tail and has a larger grid)."""
    rows = []
    for f in glob.glob(os.path.join(SRC, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "render_bwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                rows.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    if not rows:
        return None, 0
    g0 = min(g for g, _ in rows)
    if g0 == max(g for g, _ in rows):
        return None, 0          # no launch with a tail to tell them from: not an --render-bwd-alone pass of an HDR frame
    vals = [v for g, v in rows if g == g0]
    return sum(vals) / len(vals), len(vals)


if what == "pmc":
    fetch, write = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if "at::" in k or "rocclr" in k:
            continue
        f, w = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
        rows.append((k, f, w, int((2 * f + w) * 1024)))
    with open(os.path.join(DST, f"{tag}_pmc_traffic_{CFG}.csv"), "w") as o:
        o.write("kernel,FETCH_SIZE_KB_raw_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for r in rows:
            o.write(f"{r[0]},{r[1]:.1f},{r[2]:.1f},{r[3]}\n")
    d = {r[0]: r[3] for r in rows}
    pick = lambda s: next(v for k, v in d.items() if k.startswith(s))
    import subprocess
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    import hashlib
    sys.path.insert(0, ROOT)
    from bench import kernel_source_hash
    sha = kernel_source_hash(os.path.join(ROOT, "casualhdrsplat_amd", "csrc", "render.hip"))
    tpath = os.path.join(DST, "pmc_traffic.json")
    allcfg = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allcfg.update({CFG: {"render_bwd_kernel_hbm_bytes": pick("render_bwd_kernel"), "render_fwd_kernel_hbm_bytes": pick("render_fwd_kernel"),
                      "render_hip_sha256": sha,
                      "source_commit": commit + " (HEAD when the counters were folded; kernels of that tree)",
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB units); hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024, "
                                "the gfx950 FETCH_SIZE half-count correction; per-launch average; counted at the L2-fabric interface (Infinity-Cache hits included)",
                      "source": f"profiles/{tag}_pmc_traffic_{CFG}.csv"}})
    fa, nfa = render_bwd_alone("pmc_alone_FETCH_SIZE", "FETCH_SIZE")
    wa, nwa = render_bwd_alone("pmc_alone_WRITE_SIZE", "WRITE_SIZE")
    if fa is not None and wa is not None:
        allcfg[CFG]["render_bwd_kernel_tile_replay_hbm_bytes"] = int((2 * fa + wa) * 1024)
        allcfg[CFG]["tile_replay_method"] = (f"render_bwd_kernel launched by HS_BWD_RENDER alone (scripts/step_c3.py --render-bwd-alone; {nfa} / {nwa} "
                                             "launches in the FETCH / WRITE pass): no CRF-gradient tail workgroups -- to be set against 76 R' + 20 W H; "
                                             "render_bwd_kernel_hbm_bytes is the launch of a whole step, tail included (+ 24 W H algorithmic)")
    json.dump(allcfg, open(tpath, "w"), indent=1)
    if CFG != "c3":
        print(json.dumps(allcfg[CFG], indent=1))
        sys.exit(0)
    with open(os.path.join(DST, f"{tag}_pmc_sq_c3.csv"), "w") as o:
        o.write("kernel,counter,avg_per_launch\n")
        for i in range(1, 5):
            for k, cs in sorted(counters(f"pmc_sq{i}").items()):
                if k.startswith("render_"):
                    for c, v in sorted(cs.items()):
                        o.write(f"{k},{c},{v:.4e}\n")
    # VALU-busy fraction of the render kernels: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs,
    # GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
    sq = {}
    for i in range(1, 5):
        for k, cs in counters(f"pmc_sq{i}").items():
            sq.setdefault(k, {}).update(cs)
    tj = json.load(open(os.path.join(DST, "pmc_traffic.json")))
    for k, cs in sq.items():
        if k.startswith("render_") and "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
            name = k.split("<")[0]
            tj["c3"][name + "_valu_busy_frac"] = round(cs["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cs["GRBM_GUI_ACTIVE"] / 8), 4)
            tj["c3"][name + "_valu_insts_per_launch"] = cs.get("SQ_INSTS_VALU")
    tj["c3"]["valu_method"] = "SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs), profiles/%s_pmc_sq_c3.csv" % tag
    json.dump(tj, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(DST, "pmc_traffic.json")).read())
elif CFG != "c3":   # kernel stats of the bench command at another BASELINE config
    cands = glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))
    assert len(cands) == 1, cands
    shutil.copy(cands[0], os.path.join(DST, f"{tag}_bench_{CFG}_kernel_stats.csv"))
    print("copied", cands[0])
else:
    for cfg in ("c3", "c4", "c2"):
        line = open(os.path.join(SRC, f"bench_{cfg}.json")).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(DST, f"{tag}_bench_{cfg}.json"), "w").write(line + "\n")
    line = open(os.path.join(SRC, "bench_2rank_gloo_one_gpu.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, f"{tag}_bench_2rank_gloo_one_gpu.json"), "w").write(line + "\n")
    cands = glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv"))
    assert len(cands) == 1, "gpurun_out/prof_final holds several runs: delete it locally before collecting"
    st = cands[0]
    shutil.copy(st, os.path.join(DST, f"{tag}_bench_c3_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "valu_rate.txt")):
        shutil.copy(os.path.join(SRC, "valu_rate.txt"), os.path.join(DST, f"{tag}_valu_rate.txt"))
    if os.path.exists(os.path.join(SRC, "timeline.txt")):
        shutil.copy(os.path.join(SRC, "timeline.txt"), os.path.join(DST, f"{tag}_timeline_render_bwd_c3.txt"))
    for cfg in ("c3", "c4", "c2"):
        d = json.loads(open(os.path.join(DST, f"{tag}_bench_{cfg}.json")).read())
        print(cfg, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms", {k: d.get("roofline", {}).get(k) for k in ("frac", "avg_ms", "traffic")})
