#include "common.h"

namespace mtgv {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
const char* last_error_cstr() { return g_last_error.c_str(); }
}  // namespace mtgv
