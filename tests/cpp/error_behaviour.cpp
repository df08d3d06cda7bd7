// error_behaviour.cpp -- the containers keep the reference's error convention: a message on stdout and exit(1)
// (CloverMatrix4.h:779-782, CloverVector4.h:100-101).  The shape checks run before any device work, so this
// program needs no GPU.  argv[1] selects the case.
#include <CloverMatrix4.h>
#include <CloverVector4.h>

#include <cstring>

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "";
    if (!strcmp(which, "mvm")) {
        CloverMatrix4 A(128, 256);
        CloverVector4 x(128), r(128);          // x must have 256 elements
        A.mvm(x, r);
    } else if (!strcmp(which, "quantize")) {
        CloverMatrix32 A32(128, 128);
        CloverMatrix4 A(128, 256);
        A.quantize(A32);                        // shapes differ
    } else if (!strcmp(which, "mvm8")) {
        CloverMatrix4 A(128, 256);
        CloverVector8 x(256), r(256);           // r must have 128 elements
        A.mvm(x, r);
    } else if (!strcmp(which, "transpose")) {
        CloverMatrix4 A(128, 256), T(128, 256); // T must be 256 x 128
        A.transpose(T);
    } else if (!strcmp(which, "layout")) {
        // host-side layout contract: padding to 128, zeroed value padding, padding scales 1.0, scales right behind values
        CloverVector4 v(100);
        if (v.size() != 100 || v.size_pad() != 128 || v.getBytes() != 128 / 2 + 2 * 4) return 2;
        if ((char *)v.getScales() != (char *)v.getData() + 64) return 3;
        for (int i = 50; i < 64; i++) if (v.getData()[i] != 0) return 4;
        if (v.getScales()[1] != 1.0f) return 5;
        v.setBits(3, -5);
        if (v.getBits(3) != -5 || v.getBits(2) != 0) return 6;
        CloverMatrix4 M(100, 200);
        if (M.getRows() != 128 || M.getCols() != 256 || M.getBytes() != 128 * 256 / 2 + 2 * 4 * 4) return 7;
        // CloverVector8: one byte per element, scales right behind the values
        CloverVector8 v8(100);
        if (v8.size() != 100 || v8.size_pad() != 128 || v8.getBytes() != 128 + 2 * 4 || v8.getBitsLength() != 8) return 8;
        if ((char *)v8.getScales() != (char *)v8.getData() + 128) return 9;
        for (int i = 100; i < 128; i++) if (v8.getData()[i] != 0) return 10;
        if (v8.getScales()[1] != 1.0f) return 11;
        v8.setBits(5, -77);
        v8.getScales()[0] = 2.0f;
        if (v8.getBits(5) != -77 || v8.get(5) != -77 * 2.0f / 127.0f) return 12;
        printf("layout ok\n");
        return 0;
    }
    printf("not reached\n");
    return 0;
}
