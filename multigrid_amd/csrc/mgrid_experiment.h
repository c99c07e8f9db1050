// Diagnostic builds of libmgrid_hip.so (-DMGRID_EXPERIMENT, tools/build_exp.sh): phase ablation switches and clock
// stamps inside the step kernel.  The product build defines every macro below as nothing, so none of it reaches the
// library that ships; the kernel source only ever names these macros.
#ifndef MGRID_EXPERIMENT_H
#define MGRID_EXPERIMENT_H

#ifdef MGRID_EXPERIMENT
// code that exists in diagnostic builds only
#define MG_EXP(...) __VA_ARGS__
// drain memory, read the shader clock, keep it in a buffer of its own
#define MG_STAMP(k)                                                                        \
    do {                                                                                   \
        if (p.dbg_buf && !(p.dbg & 16)) {                                                  \
            __builtin_amdgcn_s_waitcnt(0);                                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (threadIdx.x == 0) p.dbg_buf[static_cast<size_t>(blockIdx.x) * 32 + (k)] = t_; \
        }                                                                                  \
    } while (0)
// light stamp: no drain, so the pipeline keeps running; the caller accumulates differences
#define MG_TICK(var)                                                                       \
    do {                                                                                   \
        if (p.dbg_buf && (p.dbg & 16)) var = __builtin_amdgcn_s_memtime();                 \
    } while (0)
#else
#define MG_EXP(...)
#define MG_STAMP(k) do { } while (0)
#define MG_TICK(var) do { } while (0)
#endif

#endif  // MGRID_EXPERIMENT_H
