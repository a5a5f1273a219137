// gp/factorized_opts_LTV.h: the factor aliases with the linear time-varying prior (LTV_GP) as the linear GP factor.
// Include this header INSTEAD of gvi_host.hpp where the reference program includes gp/factorized_opts_LTV.h.
#pragma once
#define GVI_FACTORIZED_OPTS_LTV 1
#include "gvi_host.hpp"
