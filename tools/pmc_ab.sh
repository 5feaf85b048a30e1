export TMPDIR=/tmp
D=gpurun_out/pmc_ab; mkdir -p $D
for f in 3 35; do
PB="python bench.py --model resnet101 --fusion $f --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial --no-sustained"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch$f -o x -- $PB > /dev/null 2> $D/fetch$f.err
python profiles/pmc_summary.py $D/fetch$f | head -3 > $D/fetch$f.txt
rm -rf $D/fetch$f
done
head -3 $D/fetch3.txt $D/fetch35.txt
