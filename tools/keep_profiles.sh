#!/bin/bash
# copy the summaries of gpurun_out/$TAG (tools/measure_round.sh) that are kept under profiles/, named per round
TAG=${1:-r04}
O=gpurun_out/$TAG
P=profiles
cp $O/bench.json $P/${TAG}_bench.json
cp $O/bench_alone.json $P/${TAG}_bench_alone.json
cp $O/bench_c4.json $P/${TAG}_bench_c4.json
cp $O/trace/t_kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $O/hbm_traffic.json $P/${TAG}_hbm_traffic.json
cp $O/pmc_hbm.txt $P/${TAG}_pmc_hbm.txt
cp $O/sq_counters.json $P/${TAG}_sq_counters.json
cp $O/sq_counters.txt $P/${TAG}_sq_counters.txt
cp $O/configs.jsonl $P/${TAG}_configs.jsonl
cp $O/real_text.json $P/${TAG}_real_text.json
cp $O/single_stream.jsonl $P/${TAG}_single_stream.jsonl
cp $O/single_stream_one_wave.jsonl $P/${TAG}_single_stream_one_wave.jsonl
cp $O/inflate_blocks.jsonl $P/${TAG}_inflate_blocks.jsonl
cp $O/inflate_blocks_one_wave.jsonl $P/${TAG}_inflate_blocks_one_wave.jsonl
cp $O/inflate_many.txt $P/${TAG}_inflate_many.txt
cp $O/inflate_many_kernels.txt $P/${TAG}_inflate_many_kernels.txt
cp $O/host_forms.json $P/${TAG}_host_forms.json
cp $O/torchrun.json $P/${TAG}_bench_torchrun_1proc.json
cp $O/trace_bench.json $P/${TAG}_bench_under_rocprof.json
[ -f gpurun_out/fuzz_campaign.log ] && cp gpurun_out/fuzz_campaign.log $P/${TAG}_fuzz_campaign.txt
ls -la $P | grep ${TAG}_ | wc -l
