/*
 * CloverMatrix32.h -- fp32 row-major matrix container: the input of CloverMatrix4::quantize and the
 * output of CloverMatrix4::gemm.  Mirrors the storage part of the reference's include/CloverMatrix32.h
 * (:43-73; rows and cols padded to multiples of 128 by CloverMatrix.h:48-53, contents uninitialised).
 * The class's own fp32 arithmetic -- mvm (MKL sgemv there), transpose (IPP / MKL there), quantize = copy -- is provided on the
 * HOST (clover_fp32.h): the baseline the 4-bit results are compared with, not part of the GPU path.
 */
#ifndef CLOVER_MATRIX32_H
#define CLOVER_MATRIX32_H

#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>

#include "CloverVector32.h"

class CloverMatrix32 {
protected:
    const uint64_t rows;
    const uint64_t cols;
    mutable clover_hip::Mirror mem;

public:
    CloverMatrix32(uint64_t h, uint64_t w)
        : rows(clover_hip::round_up(h, CLOVER_VECTOR_SIZE_PAD)), cols(clover_hip::round_up(w, CLOVER_VECTOR_SIZE_PAD))
    {
        mem.allocate(rows * cols * sizeof(float));
    }

    uint64_t getRows() const { return rows; }
    uint64_t getCols() const { return cols; }
    uint64_t size() const { return rows * cols; }
    uint64_t getBitsLength() const { return 32; }
    uint64_t getBytes() const { return rows * cols * sizeof(float); }

    /* Explicit residency (clover_device.h, -DCLOVER_HIP_EXPLICIT_SYNC): move the bytes NOW instead of at the next use.  toDevice(): upload
     * if the host copy is the newer one; toHost(): bring a device result back.  Optional in every build (the default build's page tracking
     * and all accessors synchronise by themselves); not in the reference, which has one copy. */
    void toDevice() const { (void)mem.dev_ro(); }
    void toHost() const { (void)mem.host_ro(); }
    float *getData() const { return reinterpret_cast<float *>(mem.host_ptr()); }      /* stays valid and current (clover_device.h) */
    float get(uint64_t i, uint64_t j) const { return reinterpret_cast<const float *>(mem.host_ro())[i * cols + j]; }
    void set(uint64_t i, uint64_t j, float v) { reinterpret_cast<float *>(mem.host_rw())[i * cols + j] = v; }
    void clear() { memset(mem.host_rw(), 0, rows * cols * sizeof(float)); }

    /* ---- fp32 arithmetic on the host (clover_fp32.h; CloverMatrix32.h:90-215) ---- */
    /* result = A * productVector; shape mismatch: message + exit(1) as the reference (:109-113) */
    void mvm(const CloverVector32 &productVector, CloverVector32 &result) const { mvm_host(productVector, result, false); }
    void mvm_parallel(const CloverVector32 &productVector, CloverVector32 &result) const { mvm_host(productVector, result, true); }
    void mvm_scalar(const CloverVector32 &productVector, CloverVector32 &result) const { mvm_host(productVector, result, false); }
    void quantize(const CloverMatrix32 &other) { memcpy(mem.host_rw(), other.mem.host_ro(), rows * cols * sizeof(float)); }
    /* other (cols x rows) = this transposed */
    void transpose(CloverMatrix32 &other) const { clover_fp32::transpose(host_ro(), rows, cols, other.host_rw(), false); }
    void transpose_parallel(CloverMatrix32 &other) const { clover_fp32::transpose(host_ro(), rows, cols, other.host_rw(), true); }
    void transpose_scalar(CloverMatrix32 &other) const { clover_fp32::transpose(host_ro(), rows, cols, other.host_rw(), false); }
    void setRandomFloats(float min_value, float max_value, uint64_t seed = 0x9E3779B97F4A7C15ull)
    {
        clover_fp32::fill_uniform(reinterpret_cast<float *>(mem.host_rw()), rows * cols, min_value, max_value, seed);
    }

    void setRandomInteger(float max_value, uint64_t seed = 0x9E3779B97F4A7C15ull)
    {
        CloverVector32 view(rows * cols, reinterpret_cast<float *>(mem.host_rw()));      /* non-owning view over the same buffer */
        view.setRandomInteger(max_value, seed);
    }

    std::string toString() const
    {
        std::stringstream sout;
        for (uint64_t i = 0; i < rows; i++) {
            for (uint64_t j = 0; j < cols; j++) sout << std::setw(7) << std::setprecision(2) << get(i, j) << " ";
            sout << ";" << std::endl;
        }
        return sout.str();
    }

    /* host access for the scalar validation twins (clover_scalar.h) */
    const float *host_ro() const { return reinterpret_cast<const float *>(mem.host_ro()); }
    float *host_rw() { return reinterpret_cast<float *>(mem.host_rw()); }
    const float *device_ro() const { return reinterpret_cast<const float *>(mem.dev_ro()); }
    float *device_wo() { return reinterpret_cast<float *>(mem.dev_wo()); }

private:
    void mvm_host(const CloverVector32 &x, CloverVector32 &result, bool team) const
    {
        if (x.size() != cols) {
            std::cout << "Can't perform MVM: " << rows << " x " << cols << " Matrix times a " << x.size() << " vector to update a "
                      << result.size() << " vector. Exiting..." << std::endl;
            exit(1);
        }
        clover_fp32::mvm_rows(host_ro(), rows, cols, x.host_ro(), result.host_rw(), team);
    }
};

#endif
