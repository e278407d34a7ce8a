// kamd_host.h -- host-side helpers shared by the translation units of libkallisto_amd.so
#pragma once
#include <string>

namespace kamd {
// records the message for kamd_last_error() and returns `code`
int fail(int code, const std::string& msg);
}
