# Counters of the generator slab kernel (tools/bench_generator.py, 640 x 35 000), one --pmc pass per counter set;
# results in gpurun_out/gen_pmc/p{1,2,3}.txt
set -e
R=$(pwd); O=$R/gpurun_out/gen_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- python3 $R/tools/bench_generator.py 640 35000 > $O/p1.log 2>&1
python3 $R/tools/pmc_dump.py $O/p1 slab > $O/p1.txt
timeout -s KILL 90 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p2 -- python3 $R/tools/bench_generator.py 640 35000 > $O/p2.log 2>&1
python3 $R/tools/pmc_dump.py $O/p2 slab > $O/p2.txt
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/p3 -- python3 $R/tools/bench_generator.py 640 35000 > $O/p3.log 2>&1
python3 $R/tools/pmc_dump.py $O/p3 slab > $O/p3.txt
find $O -name '*.csv' -size +5M -delete
