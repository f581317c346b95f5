// Transcript implementations of the host layer (see include/ceno_prover.h).
#pragma once
#include <algorithm>
#include <cstdint>

#include "../../include/ceno_prover.h"
