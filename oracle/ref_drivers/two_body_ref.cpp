/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for the Kepler two-body model
 * (SURVEY.md §8a row a17): applies mara::compute_two_body_state(full_orbital_elements_t, t) and
 * mara::compute_orbital_elements(two_body_state_t, t) of src/model_two_body.hpp to rows read from a raw f64 file.
 *
 * usage: two_body_ref <mode> <n> <in.f64> <out.f64>
 *   state     in [n][11] (pomega, tau, cmx, cmy, cmvx, cmvy, separation, total_mass, mass_ratio, eccentricity, t)   out [n][10] (body1, body2)
 *   elements  in [n][11] (body1[5], body2[5], t)     out [n][11] (the ten elements, threw)
 */
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <stdexcept>
#include <limits>
#include "model_two_body.hpp"

int main(int argc, char** argv)
{
    if (argc != 5) return 1;
    std::string mode = argv[1];
    std::size_t n = std::atol(argv[2]);
    std::vector<double> in(n * 11), out;
    FILE* f = std::fopen(argv[3], "rb");
    if (! f || std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2;
    std::fclose(f);
    for (std::size_t i = 0; i < n; ++i)
    {
        const double* x = &in[11 * i];
        if (mode == "state")
        {
            mara::full_orbital_elements_t P;
            P.pomega = x[0]; P.tau = x[1]; P.cm_position_x = x[2]; P.cm_position_y = x[3]; P.cm_velocity_x = x[4]; P.cm_velocity_y = x[5];
            P.elements.separation = x[6]; P.elements.total_mass = x[7]; P.elements.mass_ratio = x[8]; P.elements.eccentricity = x[9];
            auto s = mara::compute_two_body_state(P, x[10]);
            for (auto b : {s.body1, s.body2})
                for (double v : {b.mass, b.position_x, b.position_y, b.velocity_x, b.velocity_y}) out.push_back(v);
        }
        else
        {
            mara::two_body_state_t s;
            s.body1 = {x[0], x[1], x[2], x[3], x[4]};
            s.body2 = {x[5], x[6], x[7], x[8], x[9]};
            try
            {
                auto P = mara::compute_orbital_elements(s, x[10]);
                for (double v : {P.pomega, P.tau, P.cm_position_x, P.cm_position_y, P.cm_velocity_x, P.cm_velocity_y,
                                 P.elements.separation, P.elements.total_mass, P.elements.mass_ratio, P.elements.eccentricity}) out.push_back(v);
                out.push_back(0.0);
            }
            catch (const std::exception&)
            {
                for (int k = 0; k < 10; ++k) out.push_back(std::numeric_limits<double>::quiet_NaN());
                out.push_back(1.0);
            }
        }
    }
    FILE* g = std::fopen(argv[4], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), g);
    std::fclose(g);
    return 0;
}
