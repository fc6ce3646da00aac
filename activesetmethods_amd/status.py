"""Application return codes, values as in the reference (src/status.jl:2-22, Ipopt's codes)."""
ApplicationReturnStatus = {
    0: "Solve_Succeeded", 1: "Solved_To_Acceptable_Level", 2: "Infeasible_Problem_Detected",
    3: "Search_Direction_Becomes_Too_Small", 4: "Diverging_Iterates", 5: "User_Requested_Stop",
    6: "Feasible_Point_Found", -1: "Maximum_Iterations_Exceeded", -2: "Restoration_Failed",
    -3: "Error_In_Step_Computation", -4: "Maximum_CpuTime_Exceeded", -5: "Optimize_not_called",
    -6: "Method_not_defined", -10: "Not_Enough_Degrees_Of_Freedom", -11: "Invalid_Problem_Definition",
    -12: "Invalid_Option", -13: "Invalid_Number_Detected", -100: "Unrecoverable_Exception",
    -102: "Insufficient_Memory", -199: "Internal_Error",
}
