! run_sph_hip.f90 -- command-line front end of the Fortran host.
!
!   run_sph_hip [ic.txt] [max_steps] [final_snapshot.txt] [sph] [saves] [tend=<end time>]
!
! Options after the third argument, in any order: "sph" leaves out gas self-gravity, accretion and the boundary cull;
! "saves" writes the periodic saveN.txt files also in a step-limited run; "tend=<t>" sets the end time (default 1000).
! max_steps < 0: ingest check only -- the file is read and written back as a snapshot, no device is touched.
!
! With no arguments it behaves like the reference program (SUMMER_SPH.f90:934-955): reads
! 'disc_12000_2.txt' and runs to t = 1000 writing saveN.txt files.  With max_steps it runs that
! many steps, prints the dt sequence and (optionally) writes the final state to a snapshot.
program run_sph_hip
  use sph_hip_host
  implicit none
  character(len=512) :: filename, arg
  type(particle), allocatable :: bodies(:)
  type(sink), allocatable :: sinks(:)
  real(dp), allocatable :: dts(:)
  integer :: nsteps, k
  logical :: only_sph, with_saves
  real(dp) :: tend

  filename = 'disc_12000_2.txt'
  if (command_argument_count() >= 1) call get_command_argument(1, filename)
  call read_data_from_file(trim(filename), bodies, sinks)
  if (.not. allocated(bodies)) error stop 2

  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg)
    read(arg, *) nsteps
    only_sph = .false.
    with_saves = .false.
    tend = 1000.0_dp
    do k = 4, command_argument_count()
      call get_command_argument(k, arg)
      if (trim(arg) == 'sph') only_sph = .true.
      if (trim(arg) == 'saves') with_saves = .true.
      if (arg(1:5) == 'tend=') read(arg(6:), *) tend
    end do
    if (nsteps >= 0) then
      call simulate(bodies, sinks, end_time_in=tend, max_steps=nsteps, quiet=.true., dt_log=dts, sph_only=only_sph, saves=with_saves)
      do k = 0, ubound(dts, 1)
        write(*, '(A,I0,1X,ES25.17E3)') 'dt ', k, dts(k)
      end do
    end if
    if (command_argument_count() >= 3) then
      call get_command_argument(3, arg)
      call make_save(bodies, sinks, 0, trim(arg))
    end if
  else
    call simulate(bodies, sinks)
  end if
end program run_sph_hip
