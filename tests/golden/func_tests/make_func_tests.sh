#!/bin/bash
# Regenerates the fixtures of tests/test_gpu_func_tests.py: runs the reference's own functional test script
# (/root/reference/func_tests/runtests.sh, which writes its FASTA / FASTQ inputs itself) with the unmodified reference binary
# oracle/_ref/kallisto up to its `quant` section and keeps the inputs and the indices it built.  The expected md5s are the
# ones the reference's script holds (runtests.sh:265-304); they are listed in cases.json with the line they come from.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../../.." && pwd)
w=$(mktemp -d)
mkdir -p $w/func_tests $w/src
cp $root/oracle/_ref/kallisto $w/src/kallisto
cp /root/reference/func_tests/runtests.sh $w/
(cd $w && bash runtests.sh > run.log 2>&1 || true)
grep -c "Output OK" $w/run.log
for f in basic7.idx nonATCG.idx polyA.idx duplicates.idx small.fastq.gz simple_pair1.fastq.gz simple_pair2.fastq.gz; do cp $w/func_tests/$f $here/; done
rm -rf $w
