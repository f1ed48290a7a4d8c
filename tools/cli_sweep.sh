#!/bin/bash
# end-to-end rate (bench.py --workload cli, 1 M records per step) against the front door's knobs (round 6): host threads per stage, records
# per reader request, depth of the queues between reader / evaluation / writer
run() { echo -n "$* : "; env "$@" python bench.py --workload cli --loci 1000000 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_s']; print('%.0f records/s  read %.2f call %.2f write %.2f' % (d['value'], s['read_s'], s['call_s'], s['write_s']))"; }
run VLR_X=default
run VLR_INGEST_THREADS=8
run VLR_INGEST_THREADS=12
run VLR_CLI_CHUNK=65536
run VLR_CLI_CHUNK=131072
run VLR_CLI_QUEUE=4
run VLR_CLI_CHUNK=65536 VLR_CLI_QUEUE=4
run VLR_CLI_CHUNK=16384
