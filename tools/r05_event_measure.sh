#!/bin/bash
# Runs on the GPU box: the three measurements of VERDICT r04 task 1 (instruction table by phase, the first round of
# waves, spills) with the instrumented build.  Everything lands in gpurun_out/$1/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r05m}
mkdir -p $O
export SIGTK_AMD_LIB=$R/sigtk_amd/_variants/libsigtk_gpu_dev.so
rocprofv3 -L > $O/counters_list.txt 2>&1
EP="tools/event_phases.py --steps 3"
timeout -s KILL 600 python $EP --trace $O/trace_10000.npz > $O/phases.json 2> $O/phases.err
timeout -s KILL 600 python $EP --rna 1 > $O/phases_rna.json 2>> $O/phases.err
for n in 3072 6144 9216; do
timeout -s KILL 600 python $EP --reads $n --modes 0,1 --trace $O/trace_$n.npz > $O/phases_$n.json 2>> $O/phases.err
done
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
P3="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"
P4="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
timeout -s KILL 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc$i -- python3 $EP > $O/pmc$i.log 2>&1
i=$((i+1))
done
python tools/event_phases_pmc.py $O/phases.json $O/instruction_table.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/instruction_table.txt 2>&1
# the first round: 3 072 / 6 144 reads under the cache / translation counters that exist on this box
for n in 3072 6144; do
for P in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST" "TCC_HIT TCC_MISS TCC_REQ"; do
tag=$(echo $P | cut -d' ' -f1)
timeout -s KILL 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/fr_${n}_$tag -- python3 $EP --reads $n --modes 0 > $O/fr_${n}_$tag.log 2>&1
python tools/pmc_kernels.py $O/fr_${n}_$tag "k_event<" > $O/fr_${n}_$tag.json 2>> $O/phases.err
done
done
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
du -sh $O
cat $O/phases.json | cut -c1-1500
