// mf_kernels.hip.h -- all gfx950 kernels of the matrix-factorisation hot path (one translation unit).
#pragma once
#include "mf_common.hip.h"
#include "mf_sweep.hip.h"
#include "mf_stream.hip.h"
#include "mf_resident.hip.h"
#include "mf_recommend.hip.h"
#include "mf_collective.hip.h"
