export SGPU_TEST_HOOKS=1   # (the SGPU_* knobs these runs set are test hooks)
for v in "" "SGPU_COOP_FIRST_REACH=128" "SGPU_COOP_FIRST_REACH=256" "SGPU_COOP_FIRST_REACH=768" "SGPU_COOP_FIRST_REACH=100000" "SGPU_COOP_ITEMS_INIT=32" "SGPU_COOP_ITEMS_INIT=128" "SGPU_COOP_ITEMS_INIT=128 SGPU_COOP_MIN_ITEMS=100 SGPU_COOP_FIRST_REACH=100000" "SGPU_COOP_CHUNK_MIN=2" "SGPU_COOP_CHUNK_MIN=8" "SGPU_COOP_CHUNK_MIN=16" "SGPU_COOP_POLL=1"; do
  echo "== [$v] $(env $v python tools/latency_probe.py 8800000 2>&1 | grep -E "^nq=   1|^nq=   8|^nq=  64" | awk '{printf "%s %s us kernel | ", $1$2, $8}')"
done
