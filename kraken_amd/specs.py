"""Benchmark / fixture network specs shared by bench.py, the smoke test, the tests and the golden generator."""
BENCH_A = ('[1,48,0,1 Cr3,13,32 Do0.1,2 Mp2,2 Cr3,13,32 Do0.1,2 Mp2,2 Cr3,9,64 Do0.1,2 Mp2,2 Cr3,9,64 '
           'Do0.1,2 S1(1x0)1,3 Lbx200 Do0.1,2 Lbx200 Do0.1,2 Lbx200 Do O1c256]')
BENCH_B = '[1,48,0,1 Cr3,3,32 Gn32 Mp2,2 Cr3,3,64 Gn32 Mp2,2 S1(1x0)1,3 Lbx200 Do O1c256]'
# BENCH-A with a 3-channel input: the fixed-height / no-dewarp preprocessing case (rectangular crops of an RGB model)
BENCH_A_RGB = BENCH_A.replace('[1,48,0,1 ', '[1,48,0,3 ')
# kraken's DEFAULT recognition spec (kraken/configs/vgsl.py:102: height 120) with BENCH-A's output layer appended, as the trainer does
DEFAULT_H120 = BENCH_A.replace('[1,48,0,1 ', '[1,120,0,1 ')


def bench_codec():
    """255 single-code-point labels chr(0x100+i) <-> [i+1]."""
    return {chr(0x100 + i): [i + 1] for i in range(255)}
