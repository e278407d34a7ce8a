#!/bin/bash
# compact k-mer table on the device, C++ front-end only (no torch import: the remaining GPU budget is seconds): the reference CLI's own
# abundance.tsv of the golden cases must come out byte for byte with KAMD_TABLE_LAYOUT=compact
set -u
mkdir -p gpurun_out; O=gpurun_out/compact_check.txt; : > $O
E=kallisto_amd/kallisto_amd_quant
export KAMD_TABLE_LAYOUT=compact
T=$(mktemp -d)
mkfq(){ zcat "$1" | awk '{print "@r" NR-1; print; print "+"; s=$0; gsub(/./,"I",s); print s}' > "$2"; }
run(){ # case variant paired args...
  local c=$1 v=$2 p=$3; shift 3
  mkfq tests/golden/$c/reads_1.txt.gz $T/1.fq
  local files="$T/1.fq"
  if [ "$p" = 1 ]; then mkfq tests/golden/$c/reads_2.txt.gz $T/2.fq; files="$T/1.fq $T/2.fq"; fi
  rm -rf $T/o
  timeout 25 $E quant -i tests/golden/$c/index.idx -o $T/o --plaintext --verbose "$@" $files > $T/out.txt 2> $T/err.txt
  local rc=$?
  if [ $rc = 0 ] && cmp -s $T/o/abundance.tsv tests/golden/$c/cli_$v/abundance.tsv; then echo "$c/$v compact: IDENTICAL to the reference CLI's abundance.tsv" >> $O
  else echo "$c/$v compact: rc=$rc DIFFERENT" >> $O; tail -3 $T/err.txt >> $O; fi
  grep -i "k-mer table" $T/err.txt | head -1 >> $O
}
run ref_test_pe pe 1
run yeast_se se 0 --single -l 200 -s 20
run dlist_pe pe 1
run mosaic_pe pe_union 1 --union
run human_pe pe 1
run mosaic_pe se_nojump 0 --single -l 200 -s 25 --no-jump
cat $O
