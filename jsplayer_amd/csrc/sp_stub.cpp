// Placeholder until the ScreenPressor path lands (replaced by sp_codec.cpp).
#include <stdexcept>
#include "codec.h"
jsp_codec* jsp_make_screenpressor(int, int, int) { throw std::runtime_error("ScreenPressor path not built yet"); }
