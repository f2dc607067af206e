// Sub-program plug-in surface of the compiled host, mirroring the reference's
// `mara::sub_program_t` (src/app_subprogram.hpp:40-46): argv[0] is the sub-program
// name, the remaining arguments are key=value pairs, the return value becomes the
// process exit code. Factories are registered by hand in main.cpp, as in
// src/app_main.cpp:41-63.
#pragma once
#include <memory>
#include <string>

namespace mara {

class sub_program_t
{
public:
    virtual ~sub_program_t() {}
    virtual int main(int argc, const char* argv[]) = 0;
    virtual std::string name() const = 0;
};

} // namespace mara

std::unique_ptr<mara::sub_program_t> make_subprog_sedov();
std::unique_ptr<mara::sub_program_t> make_subprog_euler2d();
std::unique_ptr<mara::sub_program_t> make_subprog_cloud();
std::unique_ptr<mara::sub_program_t> make_subprog_binary();
