! run_sph_hip_v.f90 -- command-line front end of the variable-smoothing-length Fortran host.
!
!   run_sph_hip_v [ic10.txt] [parameters.txt | -] [max_steps] [final_snapshot.txt] [sph] [saves] [tend=<end time>]
!
! With no arguments it behaves like the reference program (Variable.f90:1168-1191): reads 'disc_20k_low_vel.txt'
! and 'parameters.txt' and runs to end_time writing saveN.txt files.  "-" (or a missing file) for the parameters
! takes the defaults of SURVEY.md 8(d).  With max_steps it runs that many steps, prints the dt sequence and
! (optionally) writes the final state to a snapshot.  Options after the fourth argument, in any order: "sph" leaves out
! gas self-gravity, accretion and the boundary cull; "saves" writes the periodic saveN.txt files also in a step-limited
! run; "tend=<t>" overrides the end time of the parameters.  max_steps < 0: ingest check only (read, write back, no device).
program run_sph_hip_v
  use sph_hip_host_v
  implicit none
  character(len=512) :: filename, pfile, arg
  type(particle), allocatable :: bodies(:)
  type(sink), allocatable :: sinks(:)
  type(param) :: params
  real(dp), allocatable :: dts(:)
  integer :: nsteps, k
  logical :: only_sph, have, with_saves

  filename = 'disc_20k_low_vel.txt'
  pfile = 'parameters.txt'
  if (command_argument_count() >= 1) call get_command_argument(1, filename)
  if (command_argument_count() >= 2) call get_command_argument(2, pfile)
  call default_params(params)
  inquire(file=trim(pfile), exist=have)
  if (trim(pfile) /= '-' .and. have) call read_params_from_file(trim(pfile), params)
  call read_data_from_file(trim(filename), bodies, sinks)
  if (.not. allocated(bodies)) error stop 2

  if (command_argument_count() >= 3) then
    call get_command_argument(3, arg)
    read(arg, *) nsteps
    only_sph = .false.
    with_saves = .false.
    do k = 5, command_argument_count()
      call get_command_argument(k, arg)
      if (trim(arg) == 'sph') only_sph = .true.
      if (trim(arg) == 'saves') with_saves = .true.
      if (arg(1:5) == 'tend=') read(arg(6:), *) params%end_time
    end do
    if (nsteps >= 0) then
      call simulate(bodies, sinks, params, max_steps=nsteps, quiet=.true., dt_log=dts, sph_only=only_sph, saves=with_saves)
      do k = 0, ubound(dts, 1)
        write(*, '(A,I0,1X,ES25.17E3)') 'dt ', k, dts(k)
      end do
    end if
    if (command_argument_count() >= 4) then
      call get_command_argument(4, arg)
      call make_save(bodies, sinks, 0, trim(arg))
    end if
  else
    call simulate(bodies, sinks, params)
  end if
end program run_sph_hip_v
