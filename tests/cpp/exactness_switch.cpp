// exactness_switch.cpp -- the headers' ONE exactness switch (include/clover_device.h) without a GPU: which mode dot() and threshold() would hand
// to the C ABI under each macro / environment / run-time setting.  Built by tests/test_exactness_switch.py with different -D flags; prints
// "dot=<0|1> threshold=<0|1>" for the start state, then the state after each run-time call.  (CLV_DOT_EXACT = 0, CLV_DOT_FAST = 1;
// CLV_THRESHOLD_FAST = 0, CLV_THRESHOLD_REFERENCE = 1.)
#include <cstdio>

#include "clover_device.h"

static void show(const char *what) { std::printf("%s dot=%d threshold=%d\n", what, clover_hip::dot_mode(), clover_hip::threshold_mode()); }

int main()
{
    show("start");
    clover_hip::set_exactness(clover_hip::FAST);
    show("set_fast");
    clover_hip::set_exactness(clover_hip::REFERENCE_BITS);
    show("set_reference");
    clover_hip::set_threshold_mode(CLV_THRESHOLD_FAST);
    show("threshold_fast_only");
    clover_hip::set_dot_mode(CLV_DOT_FAST);
    clover_hip::set_threshold_mode(CLV_THRESHOLD_REFERENCE);
    show("dot_fast_only");
    return 0;
}
