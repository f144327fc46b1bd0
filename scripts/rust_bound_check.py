#!/usr/bin/env python
"""Marker-bound checker for the Rust shim (rust/diffsol-hip), for a build container without rustc.

What rustc would reject first is not a missing method but a missing MARKER bound: `Vector: ... + Clone + Send` (diffsol-la/src/vector/mod.rs:163-177),
`Matrix: ... + Clone + Send + 'static` (matrix/mod.rs:169-170), `Context: Clone + Default`, `VectorIndex: Sized + Debug + Clone`, `LinearSolver<M>: Default`,
`VectorCommon / MatrixCommon: Sized + Debug`, and the bounds on associated types (`type Index: VectorIndex`, `type C: Context`, `type V: Vector`, ...).
This script

  1. parses every `pub trait X: A + B + ...` header (and `type N: Bound;` items) of the reference files the shim implements traits from,
     closing marker bounds over supertraits;
  2. parses the shim: struct fields, `#[derive(...)]`, `impl Trait for Type`, `unsafe impl Send/Sync for Type`, `type N = T;` inside impl blocks;
  3. for every `impl RefTrait for ShimType` asserts that each marker bound is derivable:
       Clone / Debug / Default / Copy / PartialEq  -> derived or implemented by hand,
       Send / Sync                                  -> auto-derivable from the fields (no `Rc`, no bare raw pointer, no `&T` with `T: !Sync`, `Arc<T>` only with
                                                       `T: Send + Sync`) or an explicit `unsafe impl`,
       'static                                      -> the type has no lifetime parameter,
     and that each associated type with a trait bound names a shim type that implements that trait (recursively with ITS marker bounds).

    python scripts/rust_bound_check.py [shim_src_dir]      # prints violations, exit 1 if any

It reads /root/reference (present in the build container only); tests/test_rust_shim.py skips the reference-dependent part elsewhere.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/crates"
REF_FILES = ["diffsol-la/src/vector/mod.rs", "diffsol-la/src/matrix/mod.rs", "diffsol-la/src/linear_solver/mod.rs", "diffsol-la/src/context/mod.rs",
             "diffsol-la/src/matrix/default_solver.rs", "diffsol/src/ode_equations/mod.rs", "diffsol/src/op/mod.rs", "diffsol/src/op/nonlinear_op.rs",
             "diffsol/src/op/linear_op.rs", "diffsol/src/op/constant_op.rs", "diffsol/src/ode_solver/state.rs"]
MARKERS = {"Send", "Sync", "Clone", "Debug", "Default", "Copy", "PartialEq", "Sized", "'static"}


def strip_comments(text):
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def split_top(s, sep="+"):
    """split at `sep` outside <>, (), []"""
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(s):
        c = s[i]
        if c in "<([":
            depth += 1
        elif c in ">)]" and not (c == ">" and i > 0 and s[i - 1] == "-"):
            depth -= 1
        if c == sep and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def base_name(bound):
    """`for<'a> VectorOpsByValue<&'a V, V>` -> VectorOpsByValue ; `'static` -> 'static ; `?Sized` -> None"""
    b = re.sub(r"^for\s*<[^>]*>\s*", "", bound.strip())
    if b.startswith("?"):
        return None
    if b.startswith("'"):
        return b
    m = re.match(r"(?:[A-Za-z_][\w]*::)*([A-Za-z_]\w*)", b)
    return m.group(1) if m else None


def matching_brace(text, i):
    """index just after the block that opens at text[i] == '{'"""
    depth, j = 1, i + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
    return j


def parse_reference_traits(ref=REF, files=None):
    """{trait: {"supers": [names], "markers": set, "assoc": {name: [bound names]}, "where": "file:line"}}"""
    traits = {}
    for rel in (files if files is not None else REF_FILES):
        path = os.path.join(ref, rel)
        if not os.path.exists(path):
            continue
        raw = open(path).read()
        text = strip_comments(raw)
        for m in re.finditer(r"\bpub trait\s+(\w+)\s*(<[^{;]*?>)?\s*(?::([^{]*?))?(?:\bwhere\b[^{]*)?\{", text, flags=re.S):
            name, bounds = m.group(1), m.group(3) or ""
            supers, markers = [], set()
            for b in split_top(bounds):
                n = base_name(b)
                if n is None:
                    continue
                if n in MARKERS:
                    markers.add(n)
                else:
                    supers.append(n)
            body = text[m.end() - 1:matching_brace(text, m.end() - 1)]
            assoc = {}
            for a in re.finditer(r"\btype\s+(\w+)\s*(?:<[^>]*>)?\s*:\s*([^;]+?)(?:\bwhere\b[^;]*)?;", body, flags=re.S):
                assoc[a.group(1)] = [n for n in (base_name(b) for b in split_top(a.group(2))) if n]
            lm = re.search(r"pub trait " + name + r"\b", raw)
            line = raw[:lm.start()].count("\n") + 1 if lm else 0
            traits[name] = {"supers": supers, "markers": markers, "assoc": assoc, "where": f"{rel}:{line}"}
    return traits


def closure(traits, name, seen=None):
    """marker bounds of `name` including those inherited from its supertraits: {marker: "Trait (file:line)"}"""
    seen = seen if seen is not None else set()
    out = {}
    if name in seen or name not in traits:
        return out
    seen.add(name)
    t = traits[name]
    for mk in t["markers"]:
        out.setdefault(mk, f"{name} ({t['where']})")
    for s in t["supers"]:
        for mk, src in closure(traits, s, seen).items():
            out.setdefault(mk, src)
    return out


class Shim:
    def __init__(self, src_dir):
        self.structs = {}     # name -> {"fields": [type strings], "derives": set, "lifetimes": bool, "file": f}
        self.impls = []       # (trait, type string, file, body)
        self.unsafe = set()   # (marker, type)
        self.manual = set()   # (marker trait, type)
        for fname in sorted(os.listdir(src_dir)):
            if not fname.endswith(".rs") or fname == "ffi.rs":
                continue
            text = strip_comments(open(os.path.join(src_dir, fname)).read())
            for m in re.finditer(r"((?:#\[[^\]]*\]\s*)*)pub(?:\([^)]*\))?\s+struct\s+(\w+)\s*(<[^>{(;]*>)?\s*(\{|\(|;)", text):
                attrs, name, generics, opener = m.groups()
                derives = set()
                for d in re.finditer(r"derive\(([^)]*)\)", attrs):
                    derives |= {x.strip() for x in d.group(1).split(",") if x.strip()}
                fields = []
                if opener == "{":
                    body = text[m.end():matching_brace(text, m.end() - 1) - 1]
                    for part in split_top(body, ","):
                        if ":" in part:
                            fields.append(part.split(":", 1)[1].strip())
                elif opener == "(":
                    j, depth = m.end(), 1
                    while depth:
                        depth += {"(": 1, ")": -1}.get(text[j], 0)
                        j += 1
                    for part in split_top(text[m.end():j - 1], ","):
                        fields.append(re.sub(r"^pub(?:\([^)]*\))?\s*", "", part.strip()))
                self.structs[name] = {"fields": fields, "derives": derives, "lifetimes": bool(generics and "'" in generics), "file": fname}
            for m in re.finditer(r"\bunsafe\s+impl\s+(Send|Sync)\s+for\s+(\w+)", text):
                self.unsafe.add((m.group(1), m.group(2)))
            for m in re.finditer(r"(?<!unsafe )\bimpl\s*(<[^>]*>)?\s*([\w:]+(?:<[^{]*?>)?)\s+for\s+([^{]+?)\s*\{", text):
                trait, ty = base_name(m.group(2)), m.group(3).strip()
                body = text[m.end() - 1:matching_brace(text, m.end() - 1)]
                self.impls.append((trait, ty, fname, body))
                tb = base_name(ty.lstrip("&"))
                if trait in MARKERS and tb:
                    self.manual.add((trait, tb))

    def implements(self, trait, ty_base):
        return any(t == trait and base_name(ty.lstrip("&")) == ty_base and not ty.startswith("&") for t, ty, _, _ in self.impls)

    # ---- auto traits
    def auto(self, marker, ty, stack=()):
        """(ok, reason) — is `ty` (a type expression) Send / Sync by the auto-trait rules, given the shim's structs and unsafe impls"""
        ty = ty.strip()
        if ty.startswith("*mut") or ty.startswith("*const"):
            return False, f"raw pointer `{ty}`"
        m = re.match(r"&\s*('\w+\s+)?mut\s+(.*)", ty)
        if m:
            return self.auto(marker, m.group(2), stack)  # &mut T: Send iff T: Send; Sync iff T: Sync
        m = re.match(r"&\s*('\w+\s+)?(.*)", ty)
        if m:
            return self.auto("Sync", m.group(2), stack)  # &T: Send iff T: Sync; Sync iff T: Sync
        if ty.startswith("(") and ty.endswith(")"):
            for part in split_top(ty[1:-1], ","):
                ok, why = self.auto(marker, part, stack)
                if not ok:
                    return ok, why
            return True, ""
        m = re.match(r"(?:[\w]+::)*(\w+)\s*(?:<(.*)>)?$", ty, flags=re.S)
        if not m:
            return True, ""
        name, args = m.group(1), split_top(m.group(2) or "", ",")
        args = [a for a in args if not a.startswith("'")]
        if name == "Rc":
            return False, "`Rc` is neither Send nor Sync"
        if name in ("Cell", "RefCell", "UnsafeCell"):
            if marker == "Sync":
                return False, f"`{name}` is not Sync"
            return self.auto("Send", args[0], stack) if args else (True, "")
        if name == "Arc":
            for mk in ("Send", "Sync"):
                ok, why = self.auto(mk, args[0], stack)
                if not ok:
                    return False, f"Arc<{args[0]}> needs {args[0]}: Send + Sync — {why}"
            return True, ""
        if name in ("Mutex",):
            return self.auto("Send", args[0], stack) if args else (True, "")
        if name in self.structs:
            if (marker, name) in self.unsafe:
                return True, ""
            if name in stack:
                return True, ""
            for f in self.structs[name]["fields"]:
                ok, why = self.auto(marker, f, stack + (name,))
                if not ok:
                    return False, f"{name} has a field that is not {marker}: {why}"
            return True, ""
        for a in args:  # Vec<T>, Option<T>, Box<T>, PhantomData<T>, HashMap<K, V>, ...: structural
            ok, why = self.auto(marker, a, stack)
            if not ok:
                return ok, why
        return True, ""

    def has_marker(self, marker, ty_base):
        """(ok, reason)"""
        st = self.structs.get(ty_base)
        if marker == "Sized":
            return True, ""
        if marker == "'static":
            return (not st["lifetimes"], "the type has a lifetime parameter") if st else (True, "")
        if marker in ("Send", "Sync"):
            return self.auto(marker, ty_base)
        if st and marker in st["derives"]:
            return True, ""
        if (marker, ty_base) in self.manual:
            return True, ""
        return False, f"neither derived nor implemented ({st['file'] if st else 'unknown type'})"


def check(src_dir, ref=REF):
    traits = parse_reference_traits(ref)
    shim = Shim(src_dir)
    bad, checked = [], 0
    done = set()
    log = []

    def check_impl(trait, ty_base, via):
        nonlocal checked
        if (trait, ty_base) in done or trait not in traits or ty_base not in shim.structs:
            return
        done.add((trait, ty_base))
        for mk, src in sorted(closure(traits, trait).items()):
            checked += 1
            ok, why = shim.has_marker(mk, ty_base)
            log.append(f"{ty_base}: {trait} -> {mk} [{src}]{via}: {'ok' if ok else 'MISSING'}")
            if not ok:
                bad.append(f"{ty_base}: `{trait}` requires `{mk}` (from {src}){via}: {why}")
        # associated types with trait bounds: `type X = T;` in ANY impl of `trait` (or of its supertraits) for this type
        need = {}
        stack, seen = [trait], set()
        while stack:
            t = stack.pop()
            if t in seen or t not in traits:
                continue
            seen.add(t)
            for an, bounds in traits[t]["assoc"].items():
                need.setdefault(an, set()).update(bounds)
            stack.extend(traits[t]["supers"])
        for tr, ty, fname, body in shim.impls:
            if base_name(ty.lstrip("&")) != ty_base or ty.startswith("&") or tr not in seen:
                continue
            for a in re.finditer(r"\btype\s+(\w+)\s*(?:<[^>]*>)?\s*=\s*([^;]+);", body):
                an, target = a.group(1), base_name(a.group(2).strip())
                for b in need.get(an, ()):
                    if target in shim.structs:
                        if b in MARKERS:
                            checked += 1
                            ok, why = shim.has_marker(b, target)
                            if not ok:
                                bad.append(f"{ty_base}: `{tr}::{an} = {target}` must be `{b}`: {why}")
                        elif b in traits:
                            checked += 1
                            log.append(f"{ty_base}: {tr}::{an} = {target} implements {b}: {'ok' if shim.implements(b, target) else 'MISSING'}")
                            if not shim.implements(b, target):
                                bad.append(f"{ty_base}: `{tr}::{an} = {target}` must implement `{b}` ({traits[b]['where']}) — no such impl in the shim")
                            check_impl(b, target, f" [via {ty_base}::{an}]")

    for trait, ty, fname, _ in shim.impls:
        if ty.startswith("&"):
            continue
        check_impl(trait, base_name(ty), "")
    shim.log = log
    return bad, checked, traits, shim


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    src = args[0] if args else os.path.join(ROOT, "rust", "diffsol-hip", "src")
    bad, checked, traits, shim = check(src)
    if "-v" in sys.argv:
        print("\n".join(shim.log))
    print(f"{len(traits)} reference traits parsed, {len(shim.structs)} shim types, {checked} marker / associated-type bounds checked")
    for b in bad:
        print("VIOLATION:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
