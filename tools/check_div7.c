// check_div7.c -- exhaustive check of div7() (clover_amd/csrc/common.h) against x / 7.0f (or x / 127.0f: argument 127) over all 2^32 float bit patterns.
//   gcc -O2 -fopenmp -ffp-contract=off -mfma -o /tmp/div7 tools/check_div7.c -lm && /tmp/div7     (20 s on 2 cores)
// Expected output: "mismatches 1 (...) first 0x80000000 = -0": only the sign of zero, which div7() copies from x.
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
int main(int argc, char **argv){
    const float D = argc > 1 ? (float)atof(argv[1]) : 7.0f;      /* 7 (default) or 127 */
    const float c = 1.0f/D;
    uint64_t bad=0, badn=0; uint32_t first=0; int have=0;
    #pragma omp parallel for reduction(+:bad,badn) schedule(static)
    for (uint64_t i=0;i<(1ull<<32);i++){
        uint32_t u=(uint32_t)i; float x=asf(u);
        if (isnan(x)||isinf(x)) continue;
        volatile float xv=x;
        float ref = xv/D;
        float q1 = x*c;
        float r = fmaf(-D,q1,x);
        float q2 = fmaf(r,c,q1);
        if (asu(ref)!=asu(q2)) { bad++; if (fabsf(x) >= 1e-30f) badn++; 
            #pragma omp critical
            { if(!have){first=u;have=1;} } }
    }
    printf("mismatches %llu (of which |x|>=1e-30: %llu) first 0x%08x = %g\n",(unsigned long long)bad,(unsigned long long)badn,first,asf(first));
    return 0;
}
