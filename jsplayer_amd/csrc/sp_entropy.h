// ScreenPressor entropy decoders (host stage).  Interface = EntroCoders.hx:8-24; two
// implementations: v2 range coder (RangeCoder.hx, EntroCoders.hx:31-180) and v3/v4 rANS
// (ANS.hx:5-49, EntroCoders.hx:182-313), both on top of the models in sp_models.h.
#pragma once
#include <cstdint>
#include <memory>

#include "sp_models.h"

namespace jsp::sp {

struct DecodeAbort {  // the reference would raise or never return on this stream
    const char* why;
};

class EntropyDecoder {
public:
    virtual ~EntropyDecoder() = default;
    virtual void renewI() = 0;
    virtual void begin(const uint8_t* src, size_t n, size_t pos0) = 0;
    virtual int clr(int ctx) = 0;   // one colour component; -1 = the reference's `undefined`
    // The three components of a literal colour in one call (ScreenPressor.hx:173-189 / 224-235 / 419-430): `cx` / `cx1` are the
    // caller's context halves, updated as the reference updates them after every component.  Returns the pixel, or -1 when a
    // context index leaves the tables (the reference would index past its arrays).
    virtual int64_t literal(int& cx, int& cx1, int cxshift) = 0;
    virtual int run(int ptype) = 0; // decodeN
    virtual int ptype(int prev) = 0;
    virtual int xx() = 0;
    virtual int bt() = 0;
    virtual int bn() = 0;
    virtual int sxy(int k) = 0;
    virtual int mx() = 0;
    virtual int my() = 0;
    virtual bool has_bool() const = 0;
    virtual bool flag() = 0;
    virtual bool rc_16bpp_constants() const = 0;  // differentConstantsFor16bbp
    virtual size_t consumed() const = 0;          // stream bytes read so far
};

std::unique_ptr<EntropyDecoder> make_range_decoder();
std::unique_ptr<EntropyDecoder> make_rans_decoder(int f0);

}  // namespace jsp::sp
