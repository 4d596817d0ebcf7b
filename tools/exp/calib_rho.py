import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
chrom_len, all_iv = workloads.exome_layout()
for (c0, c1, snp) in ((150, 170, False), (120, 250, True)):
    P = capi.make_params(c0, c1, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    mp = workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024, rho=-2.2)
    acc.load_model_file(mp)
    ivs = all_iv[1000:1400] + all_iv[150000:150400]
    regions = workloads.build_exome(acc, chrom_len, ivs, P, snps=snp)
    grids, scores, records = acc.score_regions(regions, capi.SCORE_SVR)
    for t in (0.05, 0.12, 0.25):
        print(c0, c1, "target", t, "rho", workloads.rho_for_exit_rate(P, grids, scores, -2.2, t))
    rho = workloads.rho_for_exit_rate(P, grids, scores, -2.2, 0.12)
    acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024, rho=rho))
    acc.upload(regions)
    acc.score_condense_all(capi.SCORE_SVR)
    em, sv = acc.download_survivors()
    print("  emitted", int(em.sum()), "dense", sum(g.count for g in grids), "frac", em.sum() / sum(g.count for g in grids))
    acc.close()
