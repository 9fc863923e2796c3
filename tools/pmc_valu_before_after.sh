# usage (on the GPU box): bash tools/pmc_valu_before_after.sh  -- VERDICT r2 item 1's evidence: SQ counters per integrate
# launch on the eight orbit views (tools/bench_views.py --orbit, one wait per frame) for the round-2 tree checked out
# and built under r02tree/ (not committed; here, before the GPU call:
#   mkdir r02tree && git archive d42c2fe | tar -x -C r02tree && make -C r02tree/yak_amd/csrc libyakhip.so &&
#   cp tools/bench_views.py r02tree/tools/ && echo r02tree >> .gitignore
# ) and for this tree, same box, same run.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/valu_ba
rm -rf $O && mkdir -p $O
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
echo "kernel time per orbit view without a profiler (library event pairs, one wait per frame), S0 / S1 beside it:"
echo -n "  round-2 tree: "; timeout -s KILL 150 python3 $R/r02tree/tools/bench_views.py "" 2>&1 | tail -1 | cut -c1-150 || exit 1
echo -n "  this tree:    "; timeout -s KILL 150 python3 $R/tools/bench_views.py "" "defer_mixed=0" 2>&1 | tail -2 | cut -c1-150 || exit 2
timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/before -- python3 $R/r02tree/tools/bench_views.py --orbit "" > $O/before.log 2>&1 || exit 3
timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/after -- python3 $R/tools/bench_views.py --orbit "" > $O/after.log 2>&1 || exit 4
timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/after1 -- python3 $R/tools/bench_views.py --orbit "defer_mixed=0" > $O/after1.log 2>&1 || exit 5
python3 - $O <<'PY'
import csv,glob,collections,sys
def table(d, title):
    disp=collections.defaultdict(dict)   # dispatch id -> {counter: value}, name
    for f in glob.glob(sys.argv[1]+"/"+d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if any(k in n for k in ("integrate","classify","checkpoint","prepass")):
                e=disp[int(r["Dispatch_Id"])]
                e["name"]=n.split("(")[0].replace("void yakhip::","").replace("yakhip::","")
                e[r["Counter_Name"]]=e.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    # a frame = a prepass launch and the kernels behind it.  tools/bench_views.py --orbit: 60 warm-up launches of the
    # wall scene, 8 untimed orbit frames, 40 timed ones, 8 more with the brick counters on: frames [-48, -8) are the timed ones
    frames=[]
    for i in sorted(disp):
        if "prepass" in disp[i]["name"]:
            frames.append([])
        if frames:
            frames[-1].append(disp[i])
    timed=frames[-48:-8]
    per=collections.defaultdict(collections.Counter); launches=collections.Counter()
    for fr in timed:
        for e in fr:
            launches[e["name"]]+=1
            for c,v in e.items():
                if c!="name": per[e["name"]][c]+=v
    print("==", title)
    print("   %d frames in the run, the 40 timed orbit frames:" % len(frames))
    tot=collections.Counter()
    for n in sorted(per):
        m={c: v/len(timed) for c,v in per[n].items()}
        print("   %-34s %3d launches: %s" % (n, launches[n], " ".join("%s=%.4g"%(c, m[c]) for c in sorted(m))))
        if "prepass" not in n: tot.update(m)
    print("   per frame, the kernels of the voxel walk (prepass left out): " + " ".join("%s=%.4g"%(c, tot[c]) for c in sorted(tot)))
    return tot
b=table("before", "BEFORE: round-2 tree (commit d42c2fe), integrate_kernel with the brick test in every workgroup")
a=table("after", "AFTER: this tree, default launch choice (two kernels on these frames: classify_kernel + integrate_list_kernel)")
a1=table("after1", "AFTER, one-kernel path of this tree forced (defer_mixed=0)")
print("SQ_INSTS_VALU per orbit frame: %.3g -> %.3g (two kernels, %.1f %%), %.3g (one kernel, %.1f %%); SQ_INSTS_SALU %.3g -> %.3g / %.3g" % (
    b["SQ_INSTS_VALU"], a["SQ_INSTS_VALU"], 100*(a["SQ_INSTS_VALU"]/b["SQ_INSTS_VALU"]-1), a1["SQ_INSTS_VALU"], 100*(a1["SQ_INSTS_VALU"]/b["SQ_INSTS_VALU"]-1),
    b["SQ_INSTS_SALU"], a["SQ_INSTS_SALU"], a1["SQ_INSTS_SALU"]))
print("(kernel times are not taken from these runs: counter collection serialises the launches)")
PY
