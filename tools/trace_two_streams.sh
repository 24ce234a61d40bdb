#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/two_stream_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o res -- python3 $ROOT/tools/two_stream_probe.py 28 > $OUT/log.txt 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import glob, sqlite3, sys
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
    print([t for t in tabs if "kernel" in t.lower() or "dispatch" in t.lower()][:12])
    try:
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        print(cols)
        rows = list(con.execute("select * from kernels order by start"))
        print(len(rows))
        idx = {c: i for i, c in enumerate(cols)}
        base = None
        for r in rows[-40:]:
            if base is None: base = r[idx["start"]]
            print({k: (r[idx[k]] - base if k in ("start", "end") else r[idx[k]]) for k in cols if k in ("start", "end", "queue_id", "stream_id", "grid_size", "name", "duration")})
    except Exception as e:
        print("err", e)
PY
