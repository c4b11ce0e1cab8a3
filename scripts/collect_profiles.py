#!/usr/bin/env python3
"""Copy the judged summaries of scripts/profile_round.sh from gpurun_out/<round>/ (scratch) into profiles/ (tracked): usage collect_profiles.py r02"""
import os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); src = os.path.join(root, "gpurun_out", R); dst = os.path.join(root, "profiles")
for f in sorted(os.listdir(src)):
    if f.endswith((".json", ".csv", "_summary.txt")) and os.path.isfile(os.path.join(src, f)):
        name = f"{R}_{f}".replace("_pmc_FETCH_SIZE", "_pmc_fetch_size").replace("_pmc_WRITE_SIZE", "_pmc_write_size")
        shutil.copy(os.path.join(src, f), os.path.join(dst, name)); print(name, os.path.getsize(os.path.join(dst, name)))
