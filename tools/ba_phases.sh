for cfg in "1000 500000 band" "1000 500000 venice" "2000 2000000 band" "1000 1000000 band"; do
  echo "== $cfg"; timeout 300 python tools/time_ba.py $cfg 2>/dev/null | grep -E "ms/solve|backsubst|schur_tiles"
done
