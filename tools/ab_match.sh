#!/bin/bash
# timing-only ablations of the matcher (results are wrong by construction)
for v in "" build/liblf_ab_SYNC.so build/liblf_ab_EPILOGUE.so build/liblf_ab_SYNC_EPILOGUE.so; do
  echo "== ${v:-product}"
  LF_MKD_LIB=${v:+$PWD/$v} python tools/bench_match.py 2>&1 | grep -E "^(65536 x 1048576|262144)"
done
