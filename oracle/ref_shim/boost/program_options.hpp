/*
 * boost/program_options.hpp — SHIM: a small from-scratch command-line parser with the slice of the
 * boost::program_options interface that the reference's main() uses (bamreadcount.cpp:434-475,641), so that the
 * reference source compiles and runs unmodified in oracle/_ref/.  TEST INFRASTRUCTURE ONLY; boost is not installed here.
 * Supported: --long value, --long=value, unambiguous long prefixes, -s value, -svalue, sticky switches (-pi),
 * positional options, default values, bool switches.
 */
#ifndef BRC_REF_SHIM_BOOST_PO_HPP
#define BRC_REF_SHIM_BOOST_PO_HPP
#include <stdint.h>

#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace boost {
namespace program_options {

struct error : std::runtime_error {
    explicit error(const std::string& w) : std::runtime_error(w) {}
};

namespace detail {
template <class T>
inline bool parse_one(const std::string& s, T& out) {
    std::istringstream is(s);
    is >> out;
    return !is.fail() && is.eof();
}
template <>
inline bool parse_one<std::string>(const std::string& s, std::string& out) { out = s; return true; }
template <>
inline bool parse_one<bool>(const std::string& s, bool& out) {
    std::string t; for (size_t i = 0; i < s.size(); ++i) t += (char)tolower((unsigned char)s[i]);
    if (t == "1" || t == "true" || t == "yes" || t == "on") { out = true; return true; }
    if (t == "0" || t == "false" || t == "no" || t == "off") { out = false; return true; }
    return false;
}
struct holder_base { virtual ~holder_base() {} };
template <class T>
struct holder : holder_base { T v; };
}  // namespace detail

class variable_value {
  public:
    std::shared_ptr<detail::holder_base> h;
    bool is_default = false;
    template <class T>
    const T& as() const {
        const detail::holder<T>* p = dynamic_cast<const detail::holder<T>*>(h.get());
        if (!p) throw error("bad any_cast of option value");
        return p->v;
    }
    bool defaulted() const { return is_default; }
    bool empty() const { return !h; }
};

class value_semantic {
  public:
    virtual ~value_semantic() {}
    virtual bool takes_arg() const = 0;
    virtual bool composing() const { return false; }
    virtual void parse(variable_value& vv, const std::string& name, const std::string* token) const = 0;
    virtual bool apply_default(variable_value& vv) const = 0;
    virtual void notify(const variable_value& vv) const = 0;
    virtual std::string describe() const { return takes_arg() ? "arg" : ""; }
};

template <class T>
class typed_value : public value_semantic {
  public:
    explicit typed_value(T* store) : store_(store) {}
    typed_value* default_value(const T& v) {
        has_default_ = true; default_ = v;
        std::ostringstream os; os << v; default_text_ = os.str();
        return this;
    }
    typed_value* implicit_value(const T& v) { has_implicit_ = true; implicit_ = v; return this; }
    typed_value* zero_tokens() { zero_tokens_ = true; return this; }
    bool takes_arg() const override { return !zero_tokens_; }
    void parse(variable_value& vv, const std::string& name, const std::string* token) const override {
        std::shared_ptr<detail::holder<T> > h(new detail::holder<T>());
        if (!token) {
            if (!has_implicit_) throw error("the required argument for option '--" + name + "' is missing");
            h->v = implicit_;
        } else if (!detail::parse_one<T>(*token, h->v)) {
            throw error("the argument ('" + *token + "') for option '--" + name + "' is invalid");
        }
        if (!vv.empty() && !vv.is_default) throw error("option '--" + name + "' cannot be specified more than once");
        vv.h = h; vv.is_default = false;
    }
    bool apply_default(variable_value& vv) const override {
        if (!has_default_) return false;
        std::shared_ptr<detail::holder<T> > h(new detail::holder<T>());
        h->v = default_; vv.h = h; vv.is_default = true;
        return true;
    }
    void notify(const variable_value& vv) const override { if (store_ && !vv.empty()) *store_ = vv.as<T>(); }
    std::string describe() const override {
        if (zero_tokens_) return "";
        return has_default_ ? "arg (=" + default_text_ + ")" : "arg";
    }

  private:
    T* store_;
    T default_ = T(), implicit_ = T();
    std::string default_text_;
    bool has_default_ = false, has_implicit_ = false, zero_tokens_ = false;
};

template <class T>
class typed_value<std::vector<T> > : public value_semantic {
  public:
    explicit typed_value(std::vector<T>* store) : store_(store) {}
    bool takes_arg() const override { return true; }
    bool composing() const override { return true; }
    void parse(variable_value& vv, const std::string& name, const std::string* token) const override {
        if (!token) throw error("the required argument for option '--" + name + "' is missing");
        std::shared_ptr<detail::holder<std::vector<T> > > h = std::dynamic_pointer_cast<detail::holder<std::vector<T> > >(vv.h);
        if (!h) { h.reset(new detail::holder<std::vector<T> >()); vv.h = h; }
        T one;
        if (!detail::parse_one<T>(*token, one)) throw error("the argument ('" + *token + "') for option '--" + name + "' is invalid");
        h->v.push_back(one); vv.is_default = false;
    }
    bool apply_default(variable_value&) const override { return false; }
    void notify(const variable_value& vv) const override { if (store_ && !vv.empty()) *store_ = vv.as<std::vector<T> >(); }

  private:
    std::vector<T>* store_;
};

template <class T>
inline typed_value<T>* value() { return new typed_value<T>(0); }
template <class T>
inline typed_value<T>* value(T* v) { return new typed_value<T>(v); }
inline typed_value<bool>* bool_switch(bool* v = 0) {
    typed_value<bool>* r = new typed_value<bool>(v);
    r->default_value(false); r->implicit_value(true); r->zero_tokens();
    return r;
}

struct option_description {
    std::string long_name, short_name, text;
    std::shared_ptr<const value_semantic> sem;
};

class options_description;
class options_description_easy_init {
  public:
    explicit options_description_easy_init(options_description* o) : owner_(o) {}
    options_description_easy_init& operator()(const char* name, const char* text);
    options_description_easy_init& operator()(const char* name, const value_semantic* s);
    options_description_easy_init& operator()(const char* name, const value_semantic* s, const char* text);

  private:
    options_description* owner_;
};

class options_description {
  public:
    options_description() {}
    explicit options_description(const std::string& caption) : caption_(caption) {}
    options_description_easy_init add_options() { return options_description_easy_init(this); }
    options_description& add(const options_description& o) {
        opts_.insert(opts_.end(), o.opts_.begin(), o.opts_.end());
        return *this;
    }
    void add_one(const char* name, const value_semantic* s, const char* text) {
        option_description d;
        std::string n(name);
        size_t c = n.find(',');
        d.long_name = n.substr(0, c);
        if (c != std::string::npos) d.short_name = n.substr(c + 1);
        d.text = text ? text : "";
        d.sem.reset(s);
        opts_.push_back(d);
    }
    const std::vector<option_description>& options() const { return opts_; }
    const option_description* find_long(const std::string& name, bool allow_prefix) const {
        const option_description* pref = 0; int npref = 0;
        for (size_t i = 0; i < opts_.size(); ++i) {
            if (opts_[i].long_name == name) return &opts_[i];
            if (allow_prefix && opts_[i].long_name.compare(0, name.size(), name) == 0) { pref = &opts_[i]; ++npref; }
        }
        if (npref > 1) throw error("option '--" + name + "' is ambiguous");
        return pref;
    }
    const option_description* find_short(char c) const {
        for (size_t i = 0; i < opts_.size(); ++i) if (opts_[i].short_name.size() == 1 && opts_[i].short_name[0] == c) return &opts_[i];
        return 0;
    }
    friend std::ostream& operator<<(std::ostream& os, const options_description& d) {
        if (!d.caption_.empty()) os << d.caption_ << ":\n";
        for (size_t i = 0; i < d.opts_.size(); ++i) {
            const option_description& o = d.opts_[i];
            std::string left = "  ";
            if (!o.short_name.empty()) left += "-" + o.short_name + " [ --" + o.long_name + " ]";
            else left += "--" + o.long_name;
            const std::string a = o.sem->describe();
            if (!a.empty()) left += " " + a;
            if (left.size() < 38) left.resize(38, ' '); else left += " ";
            os << left << o.text << "\n";
        }
        return os;
    }

  private:
    std::string caption_;
    std::vector<option_description> opts_;
};

inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const char* text) {
    typed_value<bool>* s = new typed_value<bool>(0);
    s->implicit_value(true); s->zero_tokens();
    owner_->add_one(name, s, text);
    return *this;
}
inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const value_semantic* s) {
    owner_->add_one(name, s, "");
    return *this;
}
inline options_description_easy_init& options_description_easy_init::operator()(const char* name, const value_semantic* s, const char* text) {
    owner_->add_one(name, s, text);
    return *this;
}

class positional_options_description {
  public:
    positional_options_description& add(const char* name, int max_count) {
        names_.push_back(name); counts_.push_back(max_count);
        return *this;
    }
    const std::string* name_for(size_t position) const {
        size_t acc = 0;
        for (size_t i = 0; i < names_.size(); ++i) {
            if (counts_[i] < 0) return &names_[i];
            acc += (size_t)counts_[i];
            if (position < acc) return &names_[i];
        }
        return 0;
    }

  private:
    std::vector<std::string> names_;
    std::vector<int> counts_;
};

struct parsed_option { std::string name; bool has_value; std::string value; };
struct parsed_options {
    const options_description* desc;
    std::vector<parsed_option> options;
};

class command_line_parser {
  public:
    command_line_parser(int argc, const char* const* argv) { for (int i = 1; i < argc; ++i) args_.push_back(argv[i]); }
    command_line_parser& options(const options_description& d) { desc_ = &d; return *this; }
    command_line_parser& positional(const positional_options_description& p) { pos_ = &p; return *this; }
    parsed_options run() const {
        parsed_options out; out.desc = desc_;
        size_t npos = 0; bool only_positional = false;
        for (size_t i = 0; i < args_.size(); ++i) {
            const std::string& a = args_[i];
            if (!only_positional && a == "--") { only_positional = true; continue; }
            if (!only_positional && a.size() > 2 && a[0] == '-' && a[1] == '-') {
                std::string name = a.substr(2), val; bool has = false;
                size_t eq = name.find('=');
                if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has = true; }
                const option_description* o = desc_->find_long(name, true);
                if (!o) throw error("unrecognised option '--" + name + "'");
                parsed_option p; p.name = o->long_name; p.has_value = false;
                if (o->sem->takes_arg()) {
                    if (!has) { if (i + 1 >= args_.size()) throw error("the required argument for option '--" + name + "' is missing"); val = args_[++i]; }
                    p.has_value = true; p.value = val;
                } else if (has) throw error("option '--" + name + "' does not take any arguments");
                out.options.push_back(p);
            } else if (!only_positional && a.size() > 1 && a[0] == '-') {
                for (size_t k = 1; k < a.size(); ++k) {
                    const option_description* o = desc_->find_short(a[k]);
                    if (!o) throw error(std::string("unrecognised option '-") + a[k] + "'");
                    parsed_option p; p.name = o->long_name; p.has_value = false;
                    if (o->sem->takes_arg()) {
                        std::string val = a.substr(k + 1);
                        if (val.empty()) { if (i + 1 >= args_.size()) throw error("the required argument for option '--" + o->long_name + "' is missing"); val = args_[++i]; }
                        p.has_value = true; p.value = val;
                        out.options.push_back(p);
                        break;
                    }
                    out.options.push_back(p);
                }
            } else {
                const std::string* n = pos_ ? pos_->name_for(npos) : 0;
                if (!n) throw error("too many positional options have been specified on the command line");
                ++npos;
                parsed_option p; p.name = *n; p.has_value = true; p.value = a;
                out.options.push_back(p);
            }
        }
        return out;
    }

  private:
    std::vector<std::string> args_;
    const options_description* desc_ = 0;
    const positional_options_description* pos_ = 0;
};

class variables_map {
  public:
    size_t count(const std::string& name) const { return m_.count(name); }
    const variable_value& operator[](const std::string& name) const {
        static const variable_value empty;
        std::map<std::string, variable_value>::const_iterator it = m_.find(name);
        return it == m_.end() ? empty : it->second;
    }
    std::map<std::string, variable_value> m_;
    const options_description* desc_ = 0;
};

inline void store(const parsed_options& po, variables_map& vm) {
    vm.desc_ = po.desc;
    for (size_t i = 0; i < po.options.size(); ++i) {
        const parsed_option& p = po.options[i];
        const option_description* o = po.desc->find_long(p.name, false);
        if (!o) throw error("unrecognised option '--" + p.name + "'");
        o->sem->parse(vm.m_[p.name], p.name, p.has_value ? &p.value : 0);
    }
    const std::vector<option_description>& all = po.desc->options();
    for (size_t i = 0; i < all.size(); ++i) {
        if (vm.m_.count(all[i].long_name)) continue;
        variable_value vv;
        if (all[i].sem->apply_default(vv)) vm.m_[all[i].long_name] = vv;
    }
}
inline void notify(variables_map& vm) {
    if (!vm.desc_) return;
    const std::vector<option_description>& all = vm.desc_->options();
    for (size_t i = 0; i < all.size(); ++i) {
        std::map<std::string, variable_value>::const_iterator it = vm.m_.find(all[i].long_name);
        if (it != vm.m_.end()) all[i].sem->notify(it->second);
    }
}

}  // namespace program_options
}  // namespace boost
#endif
