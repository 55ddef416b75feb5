/* placeholder until the fp16 restatement lands */ int ora_fp16_available(void) { return 0; }
