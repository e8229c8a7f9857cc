bash tools/_r04_run2.sh 2>&1 | grep -E "passed|failed|fe_rb_a|L1_offset|rc_rb|HRconv|L1_om|upconv2"
bash tools/_r04_run10.sh | grep -E "lifetime|chunk 3|epilogue|wave"
