# minimal undefined-global check: names loaded at module/function level that are neither builtins, imports, assignments nor params
import ast, builtins, sys
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read())
    defined = set(dir(builtins))
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names: defined.add((a.asname or a.name).split('.')[0])
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            defined.add(n.name)
            if not isinstance(n, ast.ClassDef):
                for a in n.args.args + n.args.kwonlyargs + n.args.posonlyargs: defined.add(a.arg)
                if n.args.vararg: defined.add(n.args.vararg.arg)
                if n.args.kwarg: defined.add(n.args.kwarg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)): defined.add(n.id)
        elif isinstance(n, ast.arg): defined.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name: defined.add(n.name)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in defined:
            print(f"{path}:{n.lineno}: undefined name {n.id}")
