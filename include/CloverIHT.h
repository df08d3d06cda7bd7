/*
 * CloverIHT.h -- the two application loops that call the 4-bit hot path in the reference:
 * quantized Iterative Hard Thresholding and quantized Gradient Descent
 * (test/performance/01_measure.h:923-946 and :999-1021).
 *
 * Same function names, argument order and step sequence as the reference's templates.  With the containers
 * of this directory every step is a kernel on the device mirrors and nothing is copied back between steps:
 * Phi, PhiT, x, y and the temporaries stay in HBM for the whole loop (the reference re-streams them from
 * DRAM through the caches every iteration).
 */
#ifndef CLOVER_IHT_H
#define CLOVER_IHT_H

#include "CloverMatrix4.h"
#include "CloverVector4.h"

/* Generic forms, for any container pair with the reference's method names (the five steps of 01_measure.h:930-944):
 * residual r = y - Phi x, gradient g = Phi' r, step x += mu g, then (IHT only) keep the K largest entries. */
template <class QMatrix, class QVector>
inline void Q_IHT(QMatrix &Phi, QMatrix &PhiT, QVector &x, QVector &y, QVector &t1, QVector &t2, QVector &t3,
                  const uint64_t iterations, const uint64_t K, const float mu)
{
    x.clear();
    for (uint64_t it = 0; it < iterations; ++it) {
        Phi.mvm_parallel(x, t1);
        y.scaleAndAdd_parallel(t1, -1.0f, t2);
        PhiT.mvm_parallel(t2, t3);
        x.scaleAndAdd_parallel(t3, mu);
        x.threshold_parallel(K);
    }
}

template <class QMatrix, class QVector>
inline void Q_GD(QMatrix &Phi, QMatrix &PhiT, QVector &x, QVector &y, QVector &t1, QVector &t2, QVector &t3,
                 const uint64_t iterations, const float mu)
{
    x.clear();
    for (uint64_t it = 0; it < iterations; ++it) {
        Phi.mvm_parallel(x, t1);
        y.scaleAndAdd_parallel(t1, -1.0f, t2);
        PhiT.mvm_parallel(t2, t3);
        x.scaleAndAdd_parallel(t3, mu);
    }
}


/* The 4-bit containers of this directory run the whole loop in ONE call when rounding is deterministic (CloverMatrix4::iht_loop ->
 * clm4_iht: a persistent launch with Phi and PhiT resident in LDS for the sizes the reference publishes, 9 us per iteration at
 * N = 8192), and otherwise pair every scaleAndAdd with the mvm before it (CloverMatrix4::mvm_scaleAndAdd: one launch instead of two);
 * identical results either way, so the same calls -- Q_IHT(Phi, PhiT, x, y, t1, t2, t3, ...) -- pick these overloads. */
inline void Q_IHT(CloverMatrix4 &Phi, CloverMatrix4 &PhiT, CloverVector4 &x, CloverVector4 &y, CloverVector4 &t1, CloverVector4 &t2,
                  CloverVector4 &t3, const uint64_t iterations, const uint64_t K, const float mu)
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    Phi.iht_loop(PhiT, x, y, t1, t2, t3, iterations, K, mu, true);
    return;
#endif
    x.clear();
    for (uint64_t i = 0; i < iterations; i += 1) {
        Phi.mvm_scaleAndAdd(x, y, -1.0f, t1, t2);      /* residual:  t1 = Phi x,  t2 = y - t1   (one launch) */
        PhiT.mvm_scaleAndAdd(t2, x, mu, t3);           /* gradient step:  t3 = Phi' t2,  x += mu t3   (one launch) */
        x.threshold_parallel(K);                       /* keep the K largest magnitudes */
    }
}

inline void Q_GD(CloverMatrix4 &Phi, CloverMatrix4 &PhiT, CloverVector4 &x, CloverVector4 &y, CloverVector4 &t1, CloverVector4 &t2,
                 CloverVector4 &t3, const uint64_t iterations, const float mu)
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    Phi.iht_loop(PhiT, x, y, t1, t2, t3, iterations, 0, mu, false);
    return;
#endif
    x.clear();
    for (uint64_t i = 0; i < iterations; i += 1) {
        Phi.mvm_scaleAndAdd(x, y, -1.0f, t1, t2);
        PhiT.mvm_scaleAndAdd(t2, x, mu, t3);
    }
}


/* CloverMatrix4 with CloverVector8 vectors: the configuration the reference measures and publishes as the 4-bit IHT / GD
 * (test/performance/02_bit04.cpp:140; "the 4-bit version uses the mixed precision MVM", doc/results/performance.txt:597-606) */
inline void Q_IHT(CloverMatrix4 &Phi, CloverMatrix4 &PhiT, CloverVector8 &x, CloverVector8 &y, CloverVector8 &t1, CloverVector8 &t2,
                  CloverVector8 &t3, const uint64_t iterations, const uint64_t K, const float mu)
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    Phi.iht_loop(PhiT, x, y, t1, t2, t3, iterations, K, mu, true);
    return;
#endif
    x.clear();
    for (uint64_t i = 0; i < iterations; i += 1) {
        Phi.mvm_scaleAndAdd(x, y, -1.0f, t1, t2);
        PhiT.mvm_scaleAndAdd(t2, x, mu, t3);
        x.threshold_parallel(K);
    }
}

inline void Q_GD(CloverMatrix4 &Phi, CloverMatrix4 &PhiT, CloverVector8 &x, CloverVector8 &y, CloverVector8 &t1, CloverVector8 &t2,
                 CloverVector8 &t3, const uint64_t iterations, const float mu)
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    Phi.iht_loop(PhiT, x, y, t1, t2, t3, iterations, 0, mu, false);
    return;
#endif
    x.clear();
    for (uint64_t i = 0; i < iterations; i += 1) {
        Phi.mvm_scaleAndAdd(x, y, -1.0f, t1, t2);
        PhiT.mvm_scaleAndAdd(t2, x, mu, t3);
    }
}

#endif
