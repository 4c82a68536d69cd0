// What the other translation units of the C ABI may ask of a handle (its definition stays in lf_mkd.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/lf_mkd.h"

int lf_mkd_internal_device(const lf_mkd *h);
hipStream_t lf_mkd_internal_stream(lf_mkd *h);
int lf_mkd_internal_fail(lf_mkd *h, int code, const std::string &msg);
