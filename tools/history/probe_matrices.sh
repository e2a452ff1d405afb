#!/bin/bash
# Round 3: can the GPU box reach the SuiteSparse collection?  A seconds-long probe; the result
# (success or the failure text) is written under gpurun_out/<dir>/matrix_probe.txt and committed under profiles/r3/.
# URLs as the reference's fetch scripts use (tests/benchmarks/matrices/get_matrices_1.sh:26-60, _2.sh:26-50);
# af_shell10 (Schenk_AFE) and Flan_1565 (Janna) are not in those scripts, collection paths from sparse.tamu.edu.
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
{
  date -u
  echo "MATRIX_DIR=${MATRIX_DIR:-<unset>}"
  ls ${MATRIX_DIR:-/nonexistent} 2>&1 | head -5
  for u in https://sparse.tamu.edu/MM/Hamm/scircuit.tar.gz https://sparse.tamu.edu/MM/Williams/webbase-1M.tar.gz \
           https://sparse.tamu.edu/MM/Schenk_AFE/af_shell10.tar.gz https://sparse.tamu.edu/MM/Janna/Flan_1565.tar.gz; do
    echo "== $u"
    timeout 12 curl -sS -I --connect-timeout 5 --max-time 10 "$u" 2>&1 | head -3
    echo "curl exit: $?"
  done
  echo "== DNS"; timeout 5 getent hosts sparse.tamu.edu; echo "getent exit: $?"
  echo "== local search for .mtx"; find / -xdev -name '*.mtx' -size +1M 2>/dev/null | head
} > $OUT/matrix_probe.txt 2>&1
tail -20 $OUT/matrix_probe.txt
