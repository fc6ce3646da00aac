"""TEST INFRASTRUCTURE (oracle): the evaluator block of the reference's MOI wrapper, restated term by term from the Julia text and kept
apart from the product's evaluator (activesetmethods_amd/moi_evaluator.py shares nothing with this file - no common flattening, no common
classes), so that it can check both the product's host evaluator and the device kernels (csrc/asm_eval_kernels.hip.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

A wrapper model is a plain dict:
    {"n": n,
     "linear_le" | "linear_ge" | "linear_eq" | "quadratic_le" | "quadratic_ge" | "quadratic_eq": [function, ...]      (MOI_wrapper.jl:14-20)
     "objective": function or None, "sense": "MIN_SENSE" | "MAX_SENSE" | "FEASIBILITY_SENSE",
     "nlp": None or {"m": rows, "pattern": [(row, col), ...] 1-based, "eval_g": f(x) -> values, "eval_jac": f(x) -> values}}
with function = {"constant": c, "affine": [(coefficient, variable), ...], "quadratic": [(coefficient, variable_1, variable_2), ...]},
variables 1-based (`VariableIndex.value`).  Every function below cites the reference lines it follows; arithmetic is plain IEEE double in the
reference's order (`function_value += coefficient * x[...]`, left-to-right products)."""

LISTS = ("linear_le", "linear_ge", "linear_eq", "quadratic_le", "quadratic_ge", "quadratic_eq")      # block order, MOI_wrapper.jl:683-689


def nlp_constraint_offset(model):
    """MOI_wrapper.jl:683-689."""
    return sum(len(model[name]) for name in LISTS)


def append_to_jacobian_sparsity(jacobian_sparsity, func, row):
    """MOI_wrapper.jl:693-712."""
    for (_, var) in func["affine"]:
        jacobian_sparsity.append((row, var))
    for (_, row_idx, col_idx) in func["quadratic"]:
        if row_idx == col_idx:
            jacobian_sparsity.append((row, row_idx))
        else:
            jacobian_sparsity.append((row, row_idx))
            jacobian_sparsity.append((row, col_idx))


def jacobian_structure(model):
    """MOI_wrapper.jl:726-746."""
    jacobian_sparsity = []
    row = 1
    for name in LISTS:
        for func in model[name]:
            append_to_jacobian_sparsity(jacobian_sparsity, func, row)
            row += 1
    if model.get("nlp") is not None:
        for (nlp_row, column) in model["nlp"]["pattern"]:
            jacobian_sparsity.append((nlp_row + row - 1, column))
    return jacobian_sparsity


def eval_function(func, x):
    """MOI_wrapper.jl:780-807 (affine and quadratic methods; x is 0-based storage of the 1-based variables)."""
    function_value = func["constant"]
    for (coefficient, var) in func["affine"]:
        function_value += coefficient * x[var - 1]
    for (coefficient, row_idx, col_idx) in func["quadratic"]:
        if row_idx == col_idx:
            function_value += 0.5 * coefficient * x[row_idx - 1] * x[col_idx - 1]
        else:
            function_value += coefficient * x[row_idx - 1] * x[col_idx - 1]
    return function_value


def objective_scale(model):
    """MOI_wrapper.jl:1037-1045."""
    return {"MIN_SENSE": 1.0, "MAX_SENSE": -1.0, "FEASIBILITY_SENSE": 0.0}[model["sense"]]


def eval_objective(model, x):
    """MOI_wrapper.jl:809-820 with the scaling of :1046-1049."""
    if model["objective"] is None:
        return 0.0
    return objective_scale(model) * eval_function(model["objective"], x)


def fill_gradient(grad, x, func):
    """MOI_wrapper.jl:827-850."""
    for j in range(len(grad)):
        grad[j] = 0.0
    for (coefficient, var) in func["affine"]:
        grad[var - 1] += coefficient
    for (coefficient, row_idx, col_idx) in func["quadratic"]:
        if row_idx == col_idx:
            grad[row_idx - 1] += coefficient * x[row_idx - 1]
        else:
            grad[row_idx - 1] += coefficient * x[col_idx - 1]
            grad[col_idx - 1] += coefficient * x[row_idx - 1]


def eval_objective_gradient(model, grad, x):
    """MOI_wrapper.jl:852-861, scaled as :1050-1054 (`rmul!(grad, objective_scale)`)."""
    if model["objective"] is not None:
        fill_gradient(grad, x, model["objective"])
    else:
        for j in range(len(grad)):
            grad[j] = 0.0
    scale = objective_scale(model)
    for j in range(len(grad)):
        grad[j] = grad[j] * scale
    return grad


def eval_constraint(model, g, x):
    """MOI_wrapper.jl:875-887."""
    row = 1
    for name in LISTS:
        for func in model[name]:
            g[row - 1] = eval_function(func, x)
            row += 1
    if model.get("nlp") is not None:
        vals = model["nlp"]["eval_g"](x)
        for k in range(len(vals)):
            g[row - 1 + k] = vals[k]
    return g


def fill_constraint_jacobian(values, start_offset, x, func):
    """MOI_wrapper.jl:889-918: returns the number of coefficients written after `start_offset` (0-based offset into `values`)."""
    num_affine_coefficients = len(func["affine"])
    for i in range(1, num_affine_coefficients + 1):
        values[start_offset + i - 1] = func["affine"][i - 1][0]
    num_quadratic_coefficients = 0
    for (coefficient, row_idx, col_idx) in func["quadratic"]:
        if row_idx == col_idx:
            values[start_offset + num_affine_coefficients + num_quadratic_coefficients] = coefficient * x[col_idx - 1]
            num_quadratic_coefficients += 1
        else:
            values[start_offset + num_affine_coefficients + num_quadratic_coefficients] = coefficient * x[col_idx - 1]
            values[start_offset + num_affine_coefficients + num_quadratic_coefficients + 1] = coefficient * x[row_idx - 1]
            num_quadratic_coefficients += 2
    return num_affine_coefficients + num_quadratic_coefficients


def eval_constraint_jacobian(model, values, x):
    """MOI_wrapper.jl:932-944."""
    offset = 0
    for name in LISTS:
        for func in model[name]:
            offset += fill_constraint_jacobian(values, offset, x, func)
    if model.get("nlp") is not None:
        vals = model["nlp"]["eval_jac"](x)
        for k in range(len(vals)):
            values[offset + k] = vals[k]
    return values
