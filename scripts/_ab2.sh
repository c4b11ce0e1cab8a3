for cfg in "MI355PT_SAH1=1" "MI355PT_SAH1=0"; do
  echo "== $cfg"
  for s in "veach_mis 1920 1080 64" "instanced_garden 1920 1080 64" "atrium 3840 2160 16"; do env $cfg timeout -k 10 200 python scripts/perf_scene.py $s 2>&1 | tail -1; done
done
