#!/bin/bash
# Large parity runs (GPU engine vs multi-threaded oracle) on the build under test, stamped with its id:
#   tools/parity_round.sh <tag>   -> gpurun_out/<tag>_parity_large.json
T=${1:-round}
O=gpurun_out/${T}_parity_large.json
BID=$(python -c "from varlociraptor_amd import engine; print(engine.build_id())")
echo "{\"build_id\": \"$BID\", \"runs\": [" > $O
first=1
for spec in "config2 200000" "config3 100000" "config4 50000" "config5 200000"; do
  set -- $spec
  line=$(python tools/large_parity.py $1 $2 2>/dev/null | grep '^{')
  [ $first = 1 ] || echo "," >> $O
  first=0
  echo "$line" >> $O
done
echo "], \"afd_fuzz\": [" >> $O
first=1
for seed in 2 3 7 11 13; do
  line=$(FUZZ_AFD=1 python tools/fuzz_scenarios.py 60 $seed 2>/dev/null | grep "^scenarios run" | sed 's/.*; //' | python -c "import sys,ast,json; print(json.dumps(dict(ast.literal_eval(sys.stdin.read().strip()), seed=$seed)))")
  [ $first = 1 ] || echo "," >> $O
  first=0
  echo "$line" >> $O
done
echo "]}" >> $O
cat $O
