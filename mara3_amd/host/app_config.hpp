// Minimal typed key=value run configuration for the compiled hosts. Same user-facing
// behaviour as the reference's mara::config_t (src/app_config.hpp): the template
// fixes each key's type through its default; `key=value` arguments update it; an
// unknown key or a value of the wrong type throws std::invalid_argument
// (src/app_config.hpp:103-136, :223-245). Not a port: one map of std::variant.
#pragma once
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <variant>

namespace mara {

class config_t
{
public:
    using value_t = std::variant<int, double, std::string>;

    config_t& item(const std::string& key, int v)                { values[key] = v; return *this; }
    config_t& item(const std::string& key, double v)             { values[key] = v; return *this; }
    config_t& item(const std::string& key, const char* v)        { values[key] = std::string(v); return *this; }
    config_t& item(const std::string& key, const std::string& v) { values[key] = v; return *this; }

    int get_int(const std::string& key) const { return std::get<int>(at(key)); }
    double get_double(const std::string& key) const { return std::get<double>(at(key)); }
    std::string get_string(const std::string& key) const { return std::get<std::string>(at(key)); }
    bool has(const std::string& key) const { return values.count(key) != 0; }
    int type_of(const std::string& key) const { return (int) at(key).index(); }         // 0 int, 1 double, 2 string
    const std::map<std::string, value_t>& items() const { return values; }

    // argv[1..] = key=value
    config_t& update(int argc, const char* argv[])
    {
        for (int n = 1; n < argc; ++n)
        {
            const std::string arg = argv[n];
            const auto eq = arg.find('=');
            if (eq == std::string::npos) throw std::invalid_argument("argument not in the form key=val: " + arg);
            const std::string key = arg.substr(0, eq), val = arg.substr(eq + 1);
            auto it = values.find(key);
            if (it == values.end()) throw std::invalid_argument("config got unknown key: " + key);
            try
            {
                std::size_t used = 0;
                switch (it->second.index())
                {
                    case 0: it->second = std::stoi(val, &used); break;
                    case 1: it->second = std::stod(val, &used); break;
                    case 2: it->second = val; used = val.size(); break;
                }
                if (used != val.size()) throw std::invalid_argument("trailing characters");
            }
            catch (const std::exception&)
            {
                throw std::invalid_argument("config got wrong data type for key " + key + ": " + val);
            }
        }
        return *this;
    }

    void pretty_print(std::FILE* out, const char* header) const
    {
        std::fprintf(out, "%s\n", std::string(52, '=').c_str());
        std::fprintf(out, "%s:\n\n", header);
        for (const auto& kv : values)
        {
            std::fprintf(out, "\t%s ", kv.first.c_str());
            for (std::size_t n = kv.first.size(); n < 24; ++n) std::fputc('.', out);
            switch (kv.second.index())
            {
                case 0: std::fprintf(out, " %d\n", std::get<int>(kv.second)); break;
                case 1: std::fprintf(out, " %g\n", std::get<double>(kv.second)); break;
                case 2: std::fprintf(out, " %s\n", std::get<std::string>(kv.second).c_str()); break;
            }
        }
        std::fprintf(out, "\n");
    }

private:
    const value_t& at(const std::string& key) const
    {
        auto it = values.find(key);
        if (it == values.end()) throw std::invalid_argument("config has no key: " + key);
        return it->second;
    }
    std::map<std::string, value_t> values;
};

} // namespace mara
