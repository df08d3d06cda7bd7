// dropin_example.cpp -- user code written against the reference's class surface (README.md:70-107 style),
// compiled against THIS repository's include/ and linked with libclover_hip.so.  Prints key=value lines that
// tests/test_gpu_cpp_dropin.py compares with the oracle / the reference's known answers.
#include <CloverIHT.h>
#include <CloverMatrix32.h>
#include <string>

#include <CloverMatrix4.h>
#include <CloverVector32.h>
#include <CloverVector4.h>

#include <cstdio>
#include <cstring>

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static void hexdump(const char *key, const int8_t *p, int n)
{
    printf("%s=", key);
    for (int i = 0; i < n; i++) printf("%02x", (unsigned)(uint8_t)p[i]);
    printf("\n");
}

int main()
{
    // ---- KAT1: the README example -----------------------------------------------------------
    {
        const int n = 128;
        CloverVector32 a_vector_32bit(n), b_vector_32bit(n);
        float *a = a_vector_32bit.getData();
        float *b = b_vector_32bit.getData();
        for (int i = 0; i < n; i += 1) { a[i] = 1; b[i] = 2; }
        CloverVector4 a_vector_4bit(128), b_vector_4bit(128);
        a_vector_4bit.quantize(a_vector_32bit);
        b_vector_4bit.quantize(b_vector_32bit);
        const float dot = a_vector_4bit.dot(b_vector_4bit);
        printf("kat1_dot=0x%08x\n", bits(dot));
        printf("kat1_dot_scalar=0x%08x\n", bits(a_vector_4bit.dot_scalar(b_vector_4bit)));
        printf("kat1_dot_parallel=%.9g\n", a_vector_4bit.dot_parallel(b_vector_4bit));
        hexdump("kat1_bytes", a_vector_4bit.getData(), 8);
        printf("kat1_scales=%g,%g\n", a_vector_4bit.getScales()[0], b_vector_4bit.getScales()[1]);
        printf("kat1_get=%g bytes=%llu\n", b_vector_4bit.get(5), (unsigned long long)a_vector_4bit.getBytes());
    }
    // ---- KAT2: quantize / dot / restore ------------------------------------------------------------
    {
        const int n = 256;
        CloverVector32 x(n), y(n), back(n);
        for (int i = 0; i < n; i++) {
            x.set(i, (float)(((37 * i) % 101) - 50) * 0.125f);
            y.set(i, (float)(((53 * i) % 89) - 44) * 0.0625f);
        }
        CloverVector4 qx(x), qy(n);
        qy.quantize_parallel(y);
        hexdump("kat2_qx", qx.getData(), 32);
        hexdump("kat2_qy", qy.getData(), 32);
        printf("kat2_dot=0x%08x\n", bits(qx.dot(qy)));
        printf("kat2_dot_scalar=0x%08x\n", bits(qx.dot_scalar(qy)));
        qx.restore(back);
        printf("kat2_restore=0x%08x,0x%08x,0x%08x,0x%08x\n", bits(back.get(0)), bits(back.get(1)), bits(back.get(2)), bits(back.get(3)));
        CloverVector4 copy(qx);                       // copy constructor keeps bytes and scales
        printf("kat2_copy_dot=0x%08x\n", bits(copy.dot(qy)));
        CloverVector4 view(n, qx.getData(), qx.getScales());   // non-owning view
        printf("kat2_view_dot=0x%08x\n", bits(view.dot(qy)));
    }
    // ---- KAT3: matrix quantize + mvm ------------------------------------------------------------------
    {
        const int M = 128, N = 256;
        CloverMatrix32 A(M, N);
        CloverVector32 x(N);
        for (int r = 0; r < M; r++)
            for (int c = 0; c < N; c++) A.set(r, c, (float)(((31 * r + 17 * c) % 23) - 11));
        for (int c = 0; c < N; c++) x.set(c, (float)(((13 * c) % 19) - 9));
        CloverMatrix4 qA(M, N);
        CloverVector4 qx(N), r(M), r2(M);
        qA.quantize(A);
        qx.quantize(x);
        qA.mvm(qx, r);
        qA.mvm_parallel(qx, r2);
        hexdump("kat3_r", r.getData(), 64);
        hexdump("kat3_r_parallel", r2.getData(), 64);
        printf("kat3_scales=0x%08x,0x%08x\n", bits(r.getScales()[0]), bits(r.getScales()[1]));
        printf("kat3_get=%.5f,%.5f,%.5f,%.5f\n", qA.get(0, 0), qA.get(0, 1), qA.get(0, 2), qA.get(0, 3));
        {   // CloverMatrix4::toString (CloverMatrix4.h:141-163): the first row as the reference prints it (setw(7), 2 decimals)
            const std::string t = qA.toString();
            printf("kat3_tostring_head=%s\n", t.substr(0, 31).c_str());
        }
        // GEMM (new): C = qA * qA^T, spot values
        CloverMatrix32 C(M, M);
        qA.gemm(qA, C);
        printf("gemm_c00=0x%08x gemm_c_1_77=0x%08x\n", bits(C.get(0, 0)), bits(C.get(1, 77)));
    }
    // ---- mixed precision: 4-bit matrix x 8-bit vector, KAT3's operands ------------------------
    {
        const int M = 128, N = 256;
        CloverMatrix32 A(M, N);
        CloverVector32 x(N), back(N);
        for (int r = 0; r < M; r++)
            for (int c = 0; c < N; c++) A.set(r, c, (float)(((31 * r + 17 * c) % 23) - 11));
        for (int c = 0; c < N; c++) x.set(c, (float)(((13 * c) % 19) - 9) * 0.37f);
        CloverMatrix4 qA(M, N);
        qA.quantize(A);
        CloverVector8 x8(x), r8(M);
        qA.mvm(x8, r8);
        hexdump("mixed_x8", x8.getData(), 64);
        hexdump("mixed_r8", r8.getData(), 128);
        printf("mixed_scales=0x%08x,0x%08x bytes8=%llu\n", bits(r8.getScales()[0]), bits(r8.getScales()[1]), (unsigned long long)x8.getBytes());
        CloverVector8 y8v(x);
        printf("mixed_dot8=0x%08x dot8_parallel=%.9g getabs=%.9g\n", bits(x8.dot(y8v)), x8.dot_parallel(y8v), x8.getAbs(1));
        x8.restore(back);
        printf("mixed_restore=0x%08x,0x%08x get=0x%08x\n", bits(back.get(1)), bits(back.get(255)), bits(x8.get(1)));
    }
    // ---- next rows: one quantized IHT-style iteration, everything device-resident ------------------------
    {
        const int M = 256, N = 512, K = 32;
        CloverMatrix32 Phi32(M, N);
        CloverVector32 x32(N), y32(M);
        Phi32.setRandomInteger(10, 7);
        x32.setRandomInteger(10, 8);
        y32.setRandomInteger(10, 9);
        CloverMatrix4 Phi(M, N), PhiT(N, M);
        CloverVector4 x(N), y(M), t1(M), t2(M), t3(N);
        Phi.quantize(Phi32);
        x.quantize(x32);
        y.quantize(y32);
        Phi.transpose(PhiT);
        Phi.mvm_parallel(x, t1);                   // t1 = Phi x
        y.scaleAndAdd_parallel(t1, -1.0f, t2);     // t2 = y - t1
        PhiT.mvm_parallel(t2, t3);                 // t3 = Phi^T t2
        x.scaleAndAdd_parallel(t3, 0.001f);        // x += mu t3
        x.threshold_parallel(K);                   // keep the K largest
        int nz = 0;
        for (int i = 0; i < N; i++) nz += x.getBits(i) != 0;
        printf("iht_nonzeros=%d\n", nz);
        printf("iht_transpose_ok=%d\n", (int)(Phi.get(3, 200) == PhiT.get(200, 3) && Phi.get(255, 0) == PhiT.get(0, 255)));
        hexdump("iht_x", x.getData(), 32);
        // the reference's loop templates (01_measure.h:923-946, 999-1021) on the same operands
        Q_IHT(Phi, PhiT, x, y, t1, t2, t3, 3, K, 0.001f);
        hexdump("qiht_x", x.getData(), 256);
        printf("qiht_scales=0x%08x,0x%08x\n", bits(x.getScales()[0]), bits(x.getScales()[7]));
        Q_GD(Phi, PhiT, x, y, t1, t2, t3, 2, 0.001f);
        hexdump("qgd_x", x.getData(), 256);
        // the same loops in the reference's published "4-bit" configuration: CloverMatrix4 with CloverVector8 vectors
        CloverVector8 x8(N), y8(y32), u1(M), u2(M), u3(N);
        Q_IHT(Phi, PhiT, x8, y8, u1, u2, u3, 3, K, 0.001f);
        hexdump("qiht8_x", x8.getData(), 512);
        printf("qiht8_scales=0x%08x,0x%08x\n", bits(x8.getScales()[0]), bits(x8.getScales()[7]));
        Q_GD(Phi, PhiT, x8, y8, u1, u2, u3, 2, 0.001f);
        hexdump("qgd8_x", x8.getData(), 512);
    }
    return 0;
}
