for d in 0.08 0.25 0.5 1.0 2.0; do python scripts/perf_scene.py atrium 3840 2160 16 "{\"detail\": $d}" 2>&1 | tail -1; done
MI355PT_BVH2=1 python scripts/perf_scene.py atrium 3840 2160 16 '{"detail": 0.08}' 2>&1 | tail -1
python scripts/perf_scene.py atrium 3840 2160 16 '{"sky_visible": false}' 2>&1 | tail -1
