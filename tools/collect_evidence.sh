#!/bin/bash
# profiles/r06_*.txt from the logs of the round's A/B runs under gpurun_out/ (each made by a scratch job of tools/perf_classes.py / tools/perf_kmc.py calls whose
# command lines are repeated in the section headers).  usage: tools/collect_evidence.sh   (from the repo root, after the gpurun calls have merged their logs)
o=gpurun_out
rows() { grep -h "launch classes:\|packed .* class:\|\"class\"" "$@" 2>/dev/null | sed -E 's/; tile 0:.*//; s/, "create_s".*//' | cut -c1-460; }
{
echo "# gibbs_single_kernel (one-cluster tiles, 168 VGPRs, 3 wavefronts / SIMD) and packed launches (several tiles per workgroup sharing a slab of LDS): measured, both opt-in"
echo "# every row: tools/perf_classes.py <S> <groups> <class> on one MI355X; ms = [first schedule, second schedule]"
echo "## r6b  S=3, 600 320 groups, whole mixture: base = hot kernel only (round 5's layout); default = single kernel, classes by kind; s4 / s4e1 / s3e4 = single kernel at 4 waves / 4 waves + one candidate per step / 3 waves + four candidates per step"
for v in base default s4 s4e1 s3e4; do echo "-- $v"; rows $o/r6b/full_S3_$v.log; done
echo "## r6d  the same batch: packed launches (pack = chosen W, T; w2 / w4_53 = forced), with and without the single kernel"
for v in packed nopack packed_nosingle base packed_w2 packed_w4_53; do echo "-- $v"; rows $o/r6d/full_S3_$v.log; done
echo "## r6f  one class ALONE at the bench's size (throughput regime): does a third wavefront per SIMD pay?"
for c in B D C; do for v in base single single_s4 packed; do echo "-- class $c, $v"; rows $o/r6f/S3_${c}_$v.log; done; done
echo "## r6g  the same with a library whose generators move no mt19937 state (-DBT_DIAG_FAKE_MT: wrong values, same control flow statistically): what the state traffic costs"
for c in B A +; do for v in base single fakemt_base fakemt_single; do echo "-- class $c, $v"; rows $o/r6g/S3_${c}_$v.log; done; done
echo "## r6g  gibbs_hot_kernel variants on the nested class: hot2o = statistics slow path out of line + two candidates per step (248 VGPRs, no spills, 2 waves); hot3 = the same at 168 VGPRs (3 waves, 67 spilled registers)"
for c in C +; do for v in hot2o hot3; do echo "-- class $c, $v"; rows $o/r6g/S3_${c}_$v.log; done; done
echo "## r06base_mem  SQ counters of the multi-variant class alone, hot kernel, one schedule under --pmc (tools/mem_counters.sh r06base B): a wavefront issues in 49 % of its cycles"
echo "##              (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES) and waits in 46 %: with two wavefronts per SIMD the SIMD's issue port is taken — VALU-bound, a third wavefront has nothing to fill"
cat $o/summ_r06base/r06base_mem_B_S3.txt 2>/dev/null
} > profiles/r06_single_kernel.txt
{
echo "# launch classes against hardware queues (GPU_MAX_HW_QUEUES; the HIP runtime's default is 4): S=3, 600 320 groups, whole mixture, tools/perf_classes.py"
echo "## r6c  q8 = GPU_MAX_HW_QUEUES=8; _133 / _123 / _132 = BT_GIBBS_KIND_CLASSES (classes per kind general,hot,single); base33 = seven fixed LDS cuts, hot kernel only"
for v in base q8_base single q8_single q8_133 q8_123 q8_132 q8_base33; do echo "-- $v"; rows $o/r6c/full_S3_$v.log; done
echo "## r6j  the final tree (class budget = the class streams that proved concurrent, cuts by dynamic programme): q4 = default queues, q8 / q16 = GPU_MAX_HW_QUEUES"
for v in q4 q8 q16; do echo "-- $v"; rows $o/r6j/S3_+_$v.log; done
echo "-- q8, two-haplotype class alone (simple_fill_unique: two samples per pass, operands of the next block requested ahead)"; rows $o/r6j/S3_A_q8.log
echo "## r6q / r6r  ring generators: chunk form (rounds 4-5, -DBT_MT_CHUNK) against block form (two buffers, next block twisted ahead cooperatively; the default), same box per pair;"
echo "##            ring1_16 / ring0_8 = BT_GIBBS_RING1=16 / BT_GIBBS_RING0=8 with the block form (32 / 16 words stay)"
for c in A B +; do for v in chunk block; do echo "-- S=3 class $c, $v"; rows $o/r6q/S3_${c}_$v.log; done; done
for v in block ring1_16 ring1_16_ring0_8; do for c in A +; do echo "-- S=3 class $c, $v"; rows $o/r6r/S3_${c}_$v.log; done; done
for v in chunk block; do echo "-- S=10, 100 352 groups, whole mixture, $v"; rows $o/r6r/S10_+_$v.log; done
} > profiles/r06_launch_classes.txt
wc -l profiles/r06_single_kernel.txt profiles/r06_launch_classes.txt
