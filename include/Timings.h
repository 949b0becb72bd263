// Timings.h -- per-stage milliseconds of the reference (include/Timings.h:4-49); same field names,
// same operators.  Converted to / from the C ABI's speck_timings by to_c() / from_c() (the flags are
// `bool` here as upstream, int32_t there: the two structs are NOT layout-compatible).
#pragma once
#include "speck_c_api.h"

struct Timings {
    bool measureAll = false;
    bool measureCompleteTime = false;
    float init = 0.0f;
    float countProducts = 0.0f;
    float loadBalanceCounting = 0.0f;
    float globalMapsCounting = 0.0f;
    float spGEMMCounting = 0.0f;
    float allocC = 0.0f;
    float loadBalanceNumeric = 0.0f;
    float globalMapsNumeric = 0.0f;
    float spGEMMNumeric = 0.0f;
    float sorting = 0.0f;
    float cleanup = 0.0f;
    float complete = 0.0f;

    void operator+=(const Timings& b)
    {
        init += b.init; countProducts += b.countProducts; loadBalanceCounting += b.loadBalanceCounting;
        globalMapsCounting += b.globalMapsCounting; spGEMMCounting += b.spGEMMCounting; allocC += b.allocC;
        loadBalanceNumeric += b.loadBalanceNumeric; globalMapsNumeric += b.globalMapsNumeric;
        spGEMMNumeric += b.spGEMMNumeric; sorting += b.sorting; cleanup += b.cleanup; complete += b.complete;
    }
    void operator/=(const float& x)
    {
        init /= x; countProducts /= x; loadBalanceCounting /= x; globalMapsCounting /= x; spGEMMCounting /= x;
        allocC /= x; loadBalanceNumeric /= x; globalMapsNumeric /= x; spGEMMNumeric /= x; sorting /= x;
        cleanup /= x; complete /= x;
    }
    speck_timings to_c() const
    {
        speck_timings t{};
        t.measureAll = measureAll;
        t.measureCompleteTime = measureCompleteTime;
        return t;
    }
    void from_c(const speck_timings& t)
    {
        init = t.init; countProducts = t.countProducts; loadBalanceCounting = t.loadBalanceCounting;
        globalMapsCounting = t.globalMapsCounting; spGEMMCounting = t.spGEMMCounting; allocC = t.allocC;
        loadBalanceNumeric = t.loadBalanceNumeric; globalMapsNumeric = t.globalMapsNumeric;
        spGEMMNumeric = t.spGEMMNumeric; sorting = t.sorting; cleanup = t.cleanup; complete = t.complete;
    }
};
